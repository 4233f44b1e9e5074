// C-ABI entry points of the vocabulary (bag-of-words) part of liborbgpu (see include/orbgpu.h).
// Reference lines are relative to /root/reference/src/ORB_SLAM2/.
#include <errno.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <new>
#include <vector>

#include "bow_internal.h"

using namespace orbgpu;

namespace {
struct VBuf {
    void *p = nullptr;
    size_t bytes = 0;
    int ensure(size_t need)
    {
        if (need <= bytes) return ORBGPU_OK;
        if (p) (void)hipFree(p);
        p = nullptr;
        bytes = 0;
        const size_t want = need + need / 4 + 256;
        if (hipMalloc(&p, want) != hipSuccess) {
            set_error("hipMalloc of %zu bytes failed", want);
            return ORBGPU_E_NOMEM;
        }
        bytes = want;
        return ORBGPU_OK;
    }
    void release()
    {
        if (p) (void)hipFree(p);
        p = nullptr;
        bytes = 0;
    }
};
}  // namespace

struct orbgpu_vocabulary {
    int device = 0;
    hipStream_t own_stream = nullptr, stream = nullptr;
    int k = 0, L = 0, scoring = 0, weighting = 0;
    int n_nodes = 0, n_words = 0, max_children = 0;
    VBuf link, desc, orig, word, weight;  // the tree in breadth-first order
    VBuf wts;                             // per-feature weights of the current call
    VBuf stage[10];                       // staging of the host entry point
};

extern "C" {

int orbgpu_vocabulary_create(int32_t device, int32_t k, int32_t L, int32_t scoring, int32_t weighting, int32_t n_nodes,
                             const int32_t *parent, const uint8_t *is_leaf, const uint8_t *desc, const double *weight,
                             orbgpu_vocabulary **out)
{
    if (!out) return ORBGPU_E_INVALID;
    *out = nullptr;
    if (n_nodes < 0 || (n_nodes > 0 && (!parent || !is_leaf || !desc || !weight))) {
        set_error("orbgpu_vocabulary_create: invalid argument");
        return ORBGPU_E_INVALID;
    }
    // the ranges loadFromTextFile accepts (Thirdparty/DBoW2/DBoW2/TemplatedVocabulary.h:1359)
    if (k < 0 || k > 20 || L < 1 || L > 10 || scoring < 0 || scoring > 5 || weighting < 0 || weighting > 3) {
        set_error("vocabulary header k=%d L=%d scoring=%d weighting=%d outside k 0..20, L 1..10, scoring 0..5, weighting 0..3", k, L,
                  scoring, weighting);
        return ORBGPU_E_INVALID;
    }
    if ((int64_t)n_nodes + 1 >= (1 << 26)) {
        set_error("vocabulary of %d nodes exceeds the supported 2^26", n_nodes);
        return ORBGPU_E_UNSUPPORTED;
    }
    const size_t N = (size_t)n_nodes + 1;  // with the root
    // children in file order (TemplatedVocabulary.h:1385-1392), words in file order of the leaves (:1408-1415)
    std::vector<int32_t> cstart(N + 1, 0), child(N), wid(N, -1);
    for (int i = 0; i < n_nodes; i++) {
        if (parent[i] < 0 || parent[i] > i) {
            set_error("vocabulary node %d names parent %d, which is not an earlier node", i + 1, parent[i]);
            return ORBGPU_E_INVALID;
        }
        cstart[(size_t)parent[i] + 1]++;
    }
    int maxch = 0;
    for (size_t j = 0; j < N; j++) {
        if (cstart[j + 1] > maxch) maxch = cstart[j + 1];
        cstart[j + 1] += cstart[j];
    }
    {
        std::vector<int32_t> cur(cstart.begin(), cstart.end() - 1);
        for (int i = 0; i < n_nodes; i++) child[(size_t)cur[parent[i]]++] = i + 1;
    }
    int n_words = 0;
    for (int i = 0; i < n_nodes; i++) {
        const bool childless = cstart[(size_t)i + 2] == cstart[(size_t)i + 1];
        if (childless != (is_leaf[i] != 0)) {
            set_error("vocabulary node %d is marked %s but has %d children", i + 1, is_leaf[i] ? "a leaf" : "an inner node",
                      cstart[(size_t)i + 2] - cstart[(size_t)i + 1]);
            return ORBGPU_E_INVALID;
        }
        if (is_leaf[i]) wid[(size_t)i + 1] = n_words++;
    }
    if (maxch > kVocMaxChildren) {
        set_error("a vocabulary node has %d children; at most %d are supported", maxch, kVocMaxChildren);
        return ORBGPU_E_UNSUPPORTED;
    }
    // breadth-first renumbering: the children of a node become contiguous
    std::vector<int32_t> order(N);  // device index -> file id
    std::vector<uint32_t> link(N);
    {
        size_t head = 0, tail = 1;
        order[0] = 0;
        while (head < tail) {
            const int32_t id = order[head];
            const int nch = cstart[(size_t)id + 1] - cstart[id];
            link[head] = ((uint32_t)tail << 6) | (uint32_t)nch;
            if (nch == 0) link[head] = 0;
            for (int c = 0; c < nch; c++) order[tail++] = child[(size_t)cstart[id] + c];
            head++;
        }
        if (tail != N) {  // cannot happen: every parent is an earlier node, so every node hangs off the root
            set_error("vocabulary tree is not connected");
            return ORBGPU_E_INVALID;
        }
    }
    std::vector<uint8_t> hdesc(N * 32, 0);
    std::vector<int32_t> hword(N);
    std::vector<double> hweight(N, 0.0);
    for (size_t j = 0; j < N; j++) {
        const int32_t id = order[j];
        hword[j] = wid[id];
        if (id > 0) {
            memcpy(&hdesc[j * 32], desc + (size_t)(id - 1) * 32, 32);
            hweight[j] = weight[id - 1];
        }
    }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) {
        set_error("no HIP device available: the vocabulary transform needs a GPU");
        return ORBGPU_E_HIP;
    }
    if (device < 0 || device >= ndev) {
        set_error("device %d out of range (%d devices)", device, ndev);
        return ORBGPU_E_INVALID;
    }
    orbgpu_vocabulary *v = new (std::nothrow) orbgpu_vocabulary();
    if (!v) return ORBGPU_E_NOMEM;
    v->device = device;
    v->k = k; v->L = L; v->scoring = scoring; v->weighting = weighting;
    v->n_nodes = n_nodes; v->n_words = n_words; v->max_children = maxch;
    int rc = ORBGPU_OK;
    do {
        if (hipSetDevice(device) != hipSuccess || hipStreamCreateWithFlags(&v->own_stream, hipStreamNonBlocking) != hipSuccess) {
            set_error("stream creation failed");
            rc = ORBGPU_E_HIP;
            break;
        }
        v->stream = v->own_stream;
        if ((rc = v->link.ensure(N * 4)) || (rc = v->desc.ensure(N * 32)) || (rc = v->orig.ensure(N * 4)) ||
            (rc = v->word.ensure(N * 4)) || (rc = v->weight.ensure(N * 8)))
            break;
        if (hipMemcpy(v->link.p, link.data(), N * 4, hipMemcpyHostToDevice) != hipSuccess ||
            hipMemcpy(v->desc.p, hdesc.data(), N * 32, hipMemcpyHostToDevice) != hipSuccess ||
            hipMemcpy(v->orig.p, order.data(), N * 4, hipMemcpyHostToDevice) != hipSuccess ||
            hipMemcpy(v->word.p, hword.data(), N * 4, hipMemcpyHostToDevice) != hipSuccess ||
            hipMemcpy(v->weight.p, hweight.data(), N * 8, hipMemcpyHostToDevice) != hipSuccess) {
            set_error("vocabulary upload failed");
            rc = ORBGPU_E_HIP;
        }
    } while (0);
    if (rc) {
        orbgpu_vocabulary_destroy(v);
        return rc;
    }
    *out = v;
    return ORBGPU_OK;
}

// TemplatedVocabulary::loadFromTextFile (Thirdparty/DBoW2/DBoW2/TemplatedVocabulary.h:1338-1424).
int orbgpu_vocabulary_load_text(int32_t device, const char *path, orbgpu_vocabulary **out)
{
    if (!out) return ORBGPU_E_INVALID;
    *out = nullptr;
    if (!path) {
        set_error("orbgpu_vocabulary_load_text: path is NULL");
        return ORBGPU_E_INVALID;
    }
    FILE *f = fopen(path, "rb");
    if (!f) {
        set_error("cannot open vocabulary file %s: %s", path, strerror(errno));
        return ORBGPU_E_INVALID;
    }
    std::vector<char> buf;
    {
        char chunk[1 << 16];
        size_t got;
        while ((got = fread(chunk, 1, sizeof(chunk), f)) > 0) buf.insert(buf.end(), chunk, chunk + got);
        fclose(f);
        buf.push_back('\0');
    }
    // one record per line, as the reference's getline loop reads them (:1378-1420)
    char *s = buf.data(), *end;
    long hdr[4] = {0, 0, 0, 0};
    std::vector<int32_t> parent;
    std::vector<uint8_t> leaf, desc;
    std::vector<double> weight;
    for (long line = 1; *s; line++) {
        char *eol = strchr(s, '\n');
        char *next = eol ? eol + 1 : s + strlen(s);
        if (eol) *eol = '\0';
        if (line == 1) {
            for (int i = 0; i < 4; i++) {
                hdr[i] = strtol(s, &end, 10);
                if (end == s) {
                    set_error("%s: the first line must hold k, L, scoring and weighting", path);
                    return ORBGPU_E_INVALID;
                }
                s = end;
            }
            s = next;
            continue;
        }
        while (*s == ' ' || *s == '\t' || *s == '\r') s++;
        if (!*s) {  // blank lines, the one after the last newline included, add no node
            s = next;
            continue;
        }
        long v[34];
        for (int i = 0; i < 34; i++) {
            v[i] = strtol(s, &end, 10);
            if (end == s) {
                set_error("%s: line %ld is not \"parent leaf 32 bytes weight\"", path, line);
                return ORBGPU_E_INVALID;
            }
            s = end;
        }
        const double w = strtod(s, &end);
        if (end == s) {
            set_error("%s: line %ld has no weight", path, line);
            return ORBGPU_E_INVALID;
        }
        parent.push_back((int32_t)v[0]);
        leaf.push_back(v[1] > 0 ? 1 : 0);  // nIsLeaf > 0, :1408
        for (int i = 0; i < 32; i++) desc.push_back((uint8_t)v[2 + i]);
        weight.push_back(w);
        s = next;
    }
    if (parent.empty() && hdr[1] == 0) {
        set_error("%s: the first line must hold k, L, scoring and weighting", path);
        return ORBGPU_E_INVALID;
    }
    return orbgpu_vocabulary_create(device, (int32_t)hdr[0], (int32_t)hdr[1], (int32_t)hdr[2], (int32_t)hdr[3], (int32_t)parent.size(),
                                    parent.data(), leaf.data(), desc.data(), weight.data(), out);
}

void orbgpu_vocabulary_destroy(orbgpu_vocabulary *v)
{
    if (!v) return;
    (void)hipSetDevice(v->device);
    if (v->stream) (void)hipStreamSynchronize(v->stream);
    v->link.release(); v->desc.release(); v->orig.release(); v->word.release(); v->weight.release(); v->wts.release();
    for (auto &b : v->stage) b.release();
    if (v->own_stream) (void)hipStreamDestroy(v->own_stream);
    delete v;
}

int orbgpu_vocabulary_set_stream(orbgpu_vocabulary *v, void *hip_stream)
{
    if (!v) return ORBGPU_E_INVALID;
    v->stream = hip_stream ? (hipStream_t)hip_stream : v->own_stream;
    return ORBGPU_OK;
}

int orbgpu_vocabulary_sync(orbgpu_vocabulary *v)
{
    if (!v) return ORBGPU_E_INVALID;
    ORBGPU_HIP_CHECK(hipStreamSynchronize(v->stream));
    return ORBGPU_OK;
}

int orbgpu_vocabulary_info(const orbgpu_vocabulary *v, int32_t *k, int32_t *L, int32_t *scoring, int32_t *weighting, int32_t *n_nodes,
                           int32_t *n_words)
{
    if (!v) return ORBGPU_E_INVALID;
    if (k) *k = v->k;
    if (L) *L = v->L;
    if (scoring) *scoring = v->scoring;
    if (weighting) *weighting = v->weighting;
    if (n_nodes) *n_nodes = v->n_nodes;
    if (n_words) *n_words = v->n_words;
    return ORBGPU_OK;
}

int orbgpu_vocabulary_transform_batch_device(orbgpu_vocabulary *v, const uint8_t *d_desc, const int32_t *d_counts, int32_t nsets,
                                             int32_t capacity, int32_t levelsup, int32_t *d_word_ids, int32_t *d_node_ids,
                                             int32_t *d_bow_ids, double *d_bow_vals, int32_t *d_n_bow, int32_t *d_fv_nodes,
                                             int32_t *d_fv_start, int32_t *d_fv_feat, int32_t *d_n_fv)
{
    if (!v || nsets < 0 || capacity < 1 || !d_desc || !d_counts || !d_word_ids || !d_node_ids || !d_bow_ids || !d_bow_vals || !d_n_bow ||
        !d_fv_nodes || !d_fv_start || !d_fv_feat || !d_n_fv) {
        set_error("orbgpu_vocabulary_transform_batch_device: invalid argument");
        return ORBGPU_E_INVALID;
    }
    if (nsets == 0) return ORBGPU_OK;
    if (capacity > kBowMaxFeatures) {
        set_error("capacity %d exceeds the %d features per set the vocabulary transform supports", capacity, kBowMaxFeatures);
        return ORBGPU_E_UNSUPPORTED;
    }
    ORBGPU_HIP_CHECK(hipSetDevice(v->device));
    hipStream_t st = v->stream;
    if (v->n_words == 0) {
        // empty(): both containers come back cleared (TemplatedVocabulary.h:1131-1137)
        ORBGPU_HIP_CHECK(hipMemsetAsync(d_n_bow, 0, (size_t)nsets * 4, st));
        ORBGPU_HIP_CHECK(hipMemsetAsync(d_n_fv, 0, (size_t)nsets * 4, st));
        ORBGPU_HIP_CHECK(hipMemsetAsync(d_fv_start, 0, (size_t)nsets * ((size_t)capacity + 1) * 4, st));
        ORBGPU_HIP_CHECK(hipMemsetAsync(d_word_ids, 0xFF, (size_t)nsets * capacity * 4, st));
        ORBGPU_HIP_CHECK(hipMemsetAsync(d_node_ids, 0, (size_t)nsets * capacity * 4, st));
        return ORBGPU_OK;
    }
    int rc;
    if ((rc = v->wts.ensure((size_t)nsets * capacity * 8))) return rc;
    BowArgs A;
    memset(&A, 0, sizeof(A));
    A.V.link = (const uint32_t *)v->link.p;
    A.V.desc = (const uint4 *)v->desc.p;
    A.V.orig = (const int32_t *)v->orig.p;
    A.V.word = (const int32_t *)v->word.p;
    A.V.weight = (const double *)v->weight.p;
    A.desc = d_desc; A.counts = d_counts; A.nsets = nsets; A.capacity = capacity;
    A.nid_level = v->L - levelsup;
    A.tf = v->weighting == 0 || v->weighting == 1;  // TF_IDF, TF (Thirdparty/DBoW2/DBoW2/BowVector.h:36-42)
    A.must = v->scoring != 5;                       // every scoring but DOT_PRODUCT normalises (ScoringObject.h:74-89)
    A.l2 = v->scoring == 1;                         // L2_NORM
    A.max_children = v->max_children;
    A.word_ids = d_word_ids; A.node_ids = d_node_ids; A.wts = (double *)v->wts.p;
    A.bow_ids = d_bow_ids; A.bow_vals = d_bow_vals; A.n_bow = d_n_bow;
    A.fv_nodes = d_fv_nodes; A.fv_start = d_fv_start; A.fv_feat = d_fv_feat; A.n_fv = d_n_fv;
    return launch_bow_transform(st, A);
}

int orbgpu_vocabulary_transform(orbgpu_vocabulary *v, const uint8_t *desc, int32_t n, int32_t levelsup, int32_t *word_ids,
                                int32_t *node_ids, int32_t *bow_ids, double *bow_vals, int32_t *n_bow, int32_t *fv_nodes,
                                int32_t *fv_start, int32_t *fv_feat, int32_t *n_fv)
{
    if (!v || n < 0 || (n > 0 && (!desc || !bow_ids || !bow_vals || !fv_nodes || !fv_feat)) || !n_bow || !fv_start || !n_fv) {
        set_error("orbgpu_vocabulary_transform: invalid argument");
        return ORBGPU_E_INVALID;
    }
    *n_bow = 0;
    *n_fv = 0;
    fv_start[0] = 0;
    if (n == 0) return ORBGPU_OK;
    ORBGPU_HIP_CHECK(hipSetDevice(v->device));
    hipStream_t st = v->stream;
    int rc;
    const size_t m = (size_t)n;
    const size_t sizes[10] = {m * 32, 4, m * 4, m * 4, m * 4, m * 8, 4, m * 4, (m + 1) * 4, m * 4};
    for (int i = 0; i < 10; i++)
        if ((rc = v->stage[i].ensure(sizes[i] + (i == 8 ? 4 : 0)))) return rc;
    int32_t *d_nfv = (int32_t *)((char *)v->stage[8].p + (m + 1) * 4);
    ORBGPU_HIP_CHECK(hipMemcpyAsync(v->stage[0].p, desc, m * 32, hipMemcpyHostToDevice, st));
    ORBGPU_HIP_CHECK(hipMemcpyAsync(v->stage[1].p, &n, 4, hipMemcpyHostToDevice, st));
    rc = orbgpu_vocabulary_transform_batch_device(v, (const uint8_t *)v->stage[0].p, (const int32_t *)v->stage[1].p, 1, n, levelsup,
                                                  (int32_t *)v->stage[2].p, (int32_t *)v->stage[3].p, (int32_t *)v->stage[4].p,
                                                  (double *)v->stage[5].p, (int32_t *)v->stage[6].p, (int32_t *)v->stage[7].p,
                                                  (int32_t *)v->stage[8].p, (int32_t *)v->stage[9].p, d_nfv);
    if (rc) return rc;
    int32_t nb = 0, nf = 0;
    ORBGPU_HIP_CHECK(hipMemcpyAsync(&nb, v->stage[6].p, 4, hipMemcpyDeviceToHost, st));
    ORBGPU_HIP_CHECK(hipMemcpyAsync(&nf, d_nfv, 4, hipMemcpyDeviceToHost, st));
    if (word_ids) ORBGPU_HIP_CHECK(hipMemcpyAsync(word_ids, v->stage[2].p, m * 4, hipMemcpyDeviceToHost, st));
    if (node_ids) ORBGPU_HIP_CHECK(hipMemcpyAsync(node_ids, v->stage[3].p, m * 4, hipMemcpyDeviceToHost, st));
    ORBGPU_HIP_CHECK(hipStreamSynchronize(st));
    if (nb > 0) {
        ORBGPU_HIP_CHECK(hipMemcpyAsync(bow_ids, v->stage[4].p, (size_t)nb * 4, hipMemcpyDeviceToHost, st));
        ORBGPU_HIP_CHECK(hipMemcpyAsync(bow_vals, v->stage[5].p, (size_t)nb * 8, hipMemcpyDeviceToHost, st));
    }
    ORBGPU_HIP_CHECK(hipMemcpyAsync(fv_start, v->stage[8].p, ((size_t)nf + 1) * 4, hipMemcpyDeviceToHost, st));
    if (nf > 0) ORBGPU_HIP_CHECK(hipMemcpyAsync(fv_nodes, v->stage[7].p, (size_t)nf * 4, hipMemcpyDeviceToHost, st));
    ORBGPU_HIP_CHECK(hipStreamSynchronize(st));
    const int32_t nkept = fv_start[nf];
    if (nkept > 0) {
        ORBGPU_HIP_CHECK(hipMemcpyAsync(fv_feat, v->stage[9].p, (size_t)nkept * 4, hipMemcpyDeviceToHost, st));
        ORBGPU_HIP_CHECK(hipStreamSynchronize(st));
    }
    *n_bow = nb;
    *n_fv = nf;
    return ORBGPU_OK;
}

}  // extern "C"
