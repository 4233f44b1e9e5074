// Argument blocks shared by the vocabulary kernels and their host launchers.
#pragma once
#include "orbgpu_internal.h"

namespace orbgpu {

constexpr int kVocMaxChildren = 32;     // children per node the descent kernel can compare in one step
constexpr int kBowMaxFeatures = 16384;  // features per descriptor set (sort keys of one set live in LDS)

// Vocabulary tree on the device.  Nodes are renumbered breadth first so that the children of a node are
// contiguous: node 0 is the root, link[j] = first child << 6 | number of children (0 for a leaf).
struct VocDev {
    const uint32_t *link;
    const uint4 *desc;     // two uint4 per node
    const int32_t *orig;   // node id in the numbering of the vocabulary file (NodeId)
    const int32_t *word;   // WordId of a leaf, -1 for inner nodes
    const double *weight;  // node weight (idf of a word; 0 = stopped)
};

struct BowArgs {
    VocDev V;
    const uint8_t *desc;      // [set][capacity][32]
    const int32_t *counts;    // [set]
    int32_t nsets, capacity;
    int32_t nid_level;        // L - levelsup
    int32_t tf, must, l2;     // weighting is TF or TF_IDF; the scoring normalises; with the L2 norm
    int32_t max_children;
    int32_t *word_ids, *node_ids;  // [set][capacity]
    double *wts;                   // [set][capacity] scratch: weight of each feature's word
    int32_t *bow_ids; double *bow_vals; int32_t *n_bow;
    int32_t *fv_nodes, *fv_start, *fv_feat, *n_fv;  // fv_start: [set][capacity + 1]
};

int launch_bow_transform(hipStream_t st, const BowArgs &A);

}  // namespace orbgpu
