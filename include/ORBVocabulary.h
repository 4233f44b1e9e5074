/*
 * ORBVocabulary.h -- host-side C++ mirror of ORB_SLAM2::ORBVocabulary
 * (reference: /root/reference/src/ORB_SLAM2/include/ORBVocabulary.h:31-32, a
 * typedef of DBoW2::TemplatedVocabulary<DBoW2::FORB::TDescriptor, DBoW2::FORB>,
 * Thirdparty/DBoW2/DBoW2/TemplatedVocabulary.h) on top of the C ABI in orbgpu.h:
 * the vocabulary tree lives on the GPU, loadFromTextFile and transform keep
 * their signatures, BowVector / FeatureVector are the same std::map types
 * (Thirdparty/DBoW2/DBoW2/BowVector.h:56-58, FeatureVector.h:21-23).
 *
 * Only what Frame::ComputeBoW / KeyFrame::ComputeBoW need is mirrored
 * (src/Frame.cc:503-510, src/KeyFrame.cc:65-76); vocabulary training and the
 * scoring objects stay with DBoW2.
 */
#ifndef ORBGPU_ORBVOCABULARY_H
#define ORBGPU_ORBVOCABULARY_H

#include <map>
#include <string>
#include <vector>

#include "ORBextractor.h"
#include "orbgpu.h"

namespace DBoW2 {
typedef unsigned int WordId;   // BowVector.h:20
typedef double WordValue;      // BowVector.h:23
typedef unsigned int NodeId;   // BowVector.h:26
class BowVector : public std::map<WordId, WordValue> {};
class FeatureVector : public std::map<NodeId, std::vector<unsigned int> > {};
}  // namespace DBoW2

namespace ORB_SLAM2 {

class ORBVocabulary {
public:
    explicit ORBVocabulary(int device = 0) : device_(device) {}
    ~ORBVocabulary() { orbgpu_vocabulary_destroy(handle_); }
    ORBVocabulary(const ORBVocabulary &) = delete;
    ORBVocabulary &operator=(const ORBVocabulary &) = delete;

    // reference: TemplatedVocabulary.h:241, 1338-1424 (called at src/System.cc:77)
    bool loadFromTextFile(const std::string &filename)
    {
        orbgpu_vocabulary *h = nullptr;
        if (orbgpu_vocabulary_load_text(device_, filename.c_str(), &h) != ORBGPU_OK) { err_ = orbgpu_last_error(); return false; }
        orbgpu_vocabulary_destroy(handle_);
        handle_ = h;
        return true;
    }

    // reference: TemplatedVocabulary.h:122-128, 185-191, 1001-1012
    bool empty() const { return size() == 0; }
    unsigned int size() const { int32_t n = 0; if (handle_) orbgpu_vocabulary_info(handle_, 0, 0, 0, 0, 0, &n); return (unsigned)n; }
    int getBranchingFactor() const { int32_t k = 0; if (handle_) orbgpu_vocabulary_info(handle_, &k, 0, 0, 0, 0, 0); return k; }
    int getDepthLevels() const { int32_t L = 0; if (handle_) orbgpu_vocabulary_info(handle_, 0, &L, 0, 0, 0, 0); return L; }

    // transform(const std::vector<TDescriptor>& features, BowVector &v, FeatureVector &fv, int levelsup)
    // (TemplatedVocabulary.h:145-146, 1127-1194) for the rows of a descriptor matrix (N x 32, 8-bit): what
    // Converter::toDescriptorVector(mDescriptors) hands to the reference.
    void transform(const orbcv::Mat &descriptors, DBoW2::BowVector &v, DBoW2::FeatureVector &fv, int levelsup)
    {
        const int n = descriptors.rows;
        if ((size_t)descriptors.step == 32 || n == 0) return transformRows(descriptors.data, n, v, fv, levelsup);
        std::vector<unsigned char> rows((size_t)n * 32);  // gather strided rows
        for (int i = 0; i < n; i++) std::memcpy(&rows[(size_t)i * 32], descriptors.data + (size_t)i * descriptors.step, 32);
        transformRows(rows.data(), n, v, fv, levelsup);
    }

    // The same from the vector of 1 x 32 rows the reference passes.
    void transform(const std::vector<orbcv::Mat> &features, DBoW2::BowVector &v, DBoW2::FeatureVector &fv, int levelsup)
    {
        std::vector<unsigned char> rows(features.size() * 32);
        for (size_t i = 0; i < features.size(); i++) std::memcpy(&rows[i * 32], features[i].data, 32);
        transformRows(rows.data(), (int)features.size(), v, fv, levelsup);
    }

    orbgpu_vocabulary *Handle() const { return handle_; }
    const std::string &LastError() const { return err_; }

private:
    void transformRows(const unsigned char *d, int n, DBoW2::BowVector &v, DBoW2::FeatureVector &fv, int levelsup)
    {
        v.clear();
        fv.clear();
        if (!handle_ || n == 0) return;
        std::vector<int32_t> bowIds(n), fvNodes(n), fvStart(n + 1), fvFeat(n);
        std::vector<double> bowVals(n);
        int32_t nBow = 0, nFv = 0;
        if (orbgpu_vocabulary_transform(handle_, d, n, levelsup, nullptr, nullptr, bowIds.data(), bowVals.data(), &nBow, fvNodes.data(),
                                        fvStart.data(), fvFeat.data(), &nFv) != ORBGPU_OK) {
            err_ = orbgpu_last_error();
            return;
        }
        for (int j = 0; j < nBow; j++) v.insert(v.end(), std::make_pair((DBoW2::WordId)bowIds[j], bowVals[j]));
        for (int j = 0; j < nFv; j++) {
            std::vector<unsigned int> &f =
                fv.insert(fv.end(), std::make_pair((DBoW2::NodeId)fvNodes[j], std::vector<unsigned int>()))->second;
            f.assign(fvFeat.begin() + fvStart[j], fvFeat.begin() + fvStart[j + 1]);
        }
    }

    int device_;
    orbgpu_vocabulary *handle_ = nullptr;
    std::string err_;
};

}  // namespace ORB_SLAM2
#endif
