/*
 * orb_oracle_frames.h -- TEST INFRASTRUCTURE, never part of the product.
 *
 * CPU restatement of the eleven search methods of ORB_SLAM2::ORBmatcher
 * (/root/reference/src/ORB_SLAM2/src/ORBmatcher.cc) on the caller's Frame / KeyFrame / MapPoint types: each
 * function walks its map points one after the other exactly as the reference does (project, gate,
 * GetFeaturesInArea, scan the candidates, update the containers) with the primitives of orb_oracle.c
 * (orc_hamming256, orc_grid_build_geom, orc_features_in_area_geom, orc_three_maxima).  Only tests include this
 * file; they compare the GPU-backed ORBmatcher of include/ORBmatcher.h with it.
 *
 * Parity unpinned: the reference holds no fixture for these functions and cannot be built here (OpenCV, DBoW2
 * absent); the cv::Mat arithmetic (3x3 float products, cv::norm, Mat::dot) is restated from OpenCV's published
 * small-matrix code paths.
 */
#ifndef ORB_ORACLE_FRAMES_H
#define ORB_ORACLE_FRAMES_H

#include <climits>
#include <cmath>
#include <cstring>
#include <set>
#include <utility>
#include <vector>

#include "orb_oracle.h"

namespace orc_frames {

// Float bounds of the Frame whose grid a KeyFrame carries (a KeyFrame itself keeps them truncated to int).
template <class KF> struct KeyFrameOrigin {
    static void get(const KF &kf, float &x, float &y) { x = (float)kf.mnMinX; y = (float)kf.mnMinY; }
};

struct P3 { float v[3]; };
struct Rt { float R[3][3]; float t[3]; };

template <class M> inline P3 col3(const M &m) { P3 p; for (int i = 0; i < 3; i++) p.v[i] = m.template at<float>(i, 0); return p; }
template <class M> inline Rt rt_of_4x4(const M &T)
{
    Rt P;
    for (int i = 0; i < 3; i++) { for (int j = 0; j < 3; j++) P.R[i][j] = T.template at<float>(i, j); P.t[i] = T.template at<float>(i, 3); }
    return P;
}
template <class MR, class MT> inline Rt rt_of(const MR &R, const MT &t)
{
    Rt P;
    for (int i = 0; i < 3; i++) { for (int j = 0; j < 3; j++) P.R[i][j] = R.template at<float>(i, j); P.t[i] = t.template at<float>(i, 0); }
    return P;
}
// cv::Mat product of a 3x3 and a 3x1 float matrix plus a 3x1: float accumulation in index order
inline P3 rt_apply(const Rt &P, const P3 &x)
{
    P3 o;
    for (int i = 0; i < 3; i++) {
        float acc = P.R[i][0] * x.v[0];
        acc = acc + P.R[i][1] * x.v[1];
        acc = acc + P.R[i][2] * x.v[2];
        o.v[i] = acc + P.t[i];
    }
    return o;
}
inline P3 rt_centre(const Rt &P)  // -R.t() * t
{
    P3 o;
    for (int i = 0; i < 3; i++) {
        float acc = P.R[0][i] * P.t[0];
        acc = acc + P.R[1][i] * P.t[1];
        acc = acc + P.R[2][i] * P.t[2];
        o.v[i] = -acc;
    }
    return o;
}
inline float l2norm(const P3 &a)  // cv::norm: double accumulator
{
    double s = 0;
    for (int i = 0; i < 3; i++) s += (double)a.v[i] * (double)a.v[i];
    return (float)std::sqrt(s);
}
inline double ddot(const P3 &a, const P3 &b)  // Mat::dot: double accumulator
{
    double s = 0;
    for (int i = 0; i < 3; i++) s += (double)a.v[i] * (double)b.v[i];
    return s;
}
inline P3 minus(const P3 &a, const P3 &b) { P3 o; for (int i = 0; i < 3; i++) o.v[i] = a.v[i] - b.v[i]; return o; }
template <class M> inline Rt rt_of_sim3(const M &S)  // :299-302
{
    double d = 0;
    for (int j = 0; j < 3; j++) d += (double)S.template at<float>(0, j) * (double)S.template at<float>(0, j);
    const float scw = (float)std::sqrt(d);
    const float k = (float)(1.0 / (double)scw);
    Rt P;
    for (int i = 0; i < 3; i++) { for (int j = 0; j < 3; j++) P.R[i][j] = S.template at<float>(i, j) * k; P.t[i] = S.template at<float>(i, 3) * k; }
    return P;
}

// GetFeaturesInArea of a Frame (src/Frame.cc:397-474) or a KeyFrame (src/KeyFrame.cc:585-624) on the oracle's grid
struct Grid {
    float g[6];
    std::vector<int32_t> cs, ci, out;
    const orc_keypoint *kps;
    int n;
    template <class KP> void build(const std::vector<KP> &keys, const float *geom6)
    {
        static_assert(sizeof(KP) == sizeof(orc_keypoint), "KeyPoint layout");
        std::memcpy(g, geom6, sizeof(g));
        kps = reinterpret_cast<const orc_keypoint *>(keys.data());
        n = (int)keys.size();
        cs.assign(ORC_GRID_COLS * ORC_GRID_ROWS + 1, 0);
        ci.assign(n > 0 ? n : 1, 0);
        out.assign(n > 0 ? n : 1, 0);
        orc_grid_build_geom(kps, n, g, cs.data(), ci.data());
    }
    std::vector<size_t> area(float x, float y, float r, int minLevel = -1, int maxLevel = -1)
    {
        const int k = orc_features_in_area_geom(kps, g, cs.data(), ci.data(), x, y, r, minLevel, maxLevel, out.data(), n);
        return std::vector<size_t>(out.begin(), out.begin() + k);
    }
};
template <class F> inline void frame_grid(const F &f, Grid &G)
{
    const float b[4] = {(float)f.mnMinX, (float)f.mnMinY, (float)f.mnMaxX, (float)f.mnMaxY};
    float g[6];
    orc_geom_from_bounds(b, g);
    G.build(f.mvKeysUn, g);
}
template <class KF> inline void keyframe_grid(const KF &kf, Grid &G)
{
    float g[6];
    KeyFrameOrigin<KF>::get(kf, g[0], g[1]);
    g[2] = (float)kf.mnMinX; g[3] = (float)kf.mnMinY; g[4] = kf.mfGridElementWidthInv; g[5] = kf.mfGridElementHeightInv;
    G.build(kf.mvKeysUn, g);
}

inline int rot_bin(float a1, float a2)  // :238-243
{
    const float factor = 1.0f / ORC_HISTO_LENGTH;
    float rot = a1 - a2;
    if (rot < 0.0) rot += 360.0f;
    int bin = (int)roundf(rot * factor);
    if (bin == ORC_HISTO_LENGTH) bin = 0;
    return bin;
}
inline void three_maxima(std::vector<int> *h, int &i1, int &i2, int &i3)
{
    int sizes[ORC_HISTO_LENGTH];
    for (int i = 0; i < ORC_HISTO_LENGTH; i++) sizes[i] = (int)h[i].size();
    orc_three_maxima(sizes, ORC_HISTO_LENGTH, &i1, &i2, &i3);
}
template <class M> inline int ham(const M &a, const uint8_t *b) { return orc_hamming256(a.ptr(0), b); }

// :45-129
template <class FrameT, class MP>
int SearchByProjection(FrameT &F, const std::vector<MP *> &vpMapPoints, const float th, float nnratio)
{
    Grid G; frame_grid(F, G);
    int nmatches = 0;
    const bool bFactor = th != 1.0;
    for (size_t iMP = 0; iMP < vpMapPoints.size(); iMP++) {
        MP *pMP = vpMapPoints[iMP];
        if (!pMP->mbTrackInView) continue;
        if (pMP->isBad()) continue;
        const int nPredictedLevel = pMP->mnTrackScaleLevel;
        float r = pMP->mTrackViewCos > 0.998 ? 2.5f : 4.0f;
        if (bFactor) r *= th;
        const std::vector<size_t> vIndices = G.area(pMP->mTrackProjX, pMP->mTrackProjY, r * F.mvScaleFactors[nPredictedLevel],
                                                    nPredictedLevel - 1, nPredictedLevel);
        if (vIndices.empty()) continue;
        const auto MPdescriptor = pMP->GetDescriptor();
        int bestDist = 256, bestLevel = -1, bestDist2 = 256, bestLevel2 = -1, bestIdx = -1;
        for (size_t k = 0; k < vIndices.size(); k++) {
            const size_t idx = vIndices[k];
            if (F.mvpMapPoints[idx])
                if (F.mvpMapPoints[idx]->Observations() > 0) continue;
            if (F.mvuRight[idx] > 0) {
                const float er = fabsf(pMP->mTrackProjXR - F.mvuRight[idx]);
                if (er > r * F.mvScaleFactors[nPredictedLevel]) continue;
            }
            const int dist = ham(MPdescriptor, F.mDescriptors.ptr((int)idx));
            if (dist < bestDist) {
                bestDist2 = bestDist; bestDist = dist; bestLevel2 = bestLevel; bestLevel = F.mvKeysUn[idx].octave; bestIdx = (int)idx;
            } else if (dist < bestDist2) {
                bestLevel2 = F.mvKeysUn[idx].octave; bestDist2 = dist;
            }
        }
        if (bestDist <= ORC_TH_HIGH) {
            if (bestLevel == bestLevel2 && bestDist > nnratio * bestDist2) continue;
            F.mvpMapPoints[bestIdx] = pMP;
            nmatches++;
        }
    }
    return nmatches;
}

// :1379-1521
template <class FrameT>
int SearchByProjection(FrameT &CurrentFrame, const FrameT &LastFrame, const float th, const bool bMono, bool checkOri)
{
    Grid G; frame_grid(CurrentFrame, G);
    int nmatches = 0;
    std::vector<int> rotHist[ORC_HISTO_LENGTH];
    const Rt cw = rt_of_4x4(CurrentFrame.mTcw), lw = rt_of_4x4(LastFrame.mTcw);
    const P3 twc = rt_centre(cw);
    const P3 tlc = rt_apply(lw, twc);
    const bool bForward = tlc.v[2] > CurrentFrame.mb && !bMono;
    const bool bBackward = -tlc.v[2] > CurrentFrame.mb && !bMono;
    for (int i = 0; i < LastFrame.N; i++) {
        auto *pMP = LastFrame.mvpMapPoints[i];
        if (!pMP) continue;
        if (LastFrame.mvbOutlier[i]) continue;
        const P3 x3Dc = rt_apply(cw, col3(pMP->GetWorldPos()));
        const float xc = x3Dc.v[0], yc = x3Dc.v[1];
        const float invzc = 1.0 / x3Dc.v[2];
        if (invzc < 0) continue;
        float u = CurrentFrame.fx * xc * invzc + CurrentFrame.cx;
        float v = CurrentFrame.fy * yc * invzc + CurrentFrame.cy;
        if (u < CurrentFrame.mnMinX || u > CurrentFrame.mnMaxX) continue;
        if (v < CurrentFrame.mnMinY || v > CurrentFrame.mnMaxY) continue;
        const int nLastOctave = LastFrame.mvKeys[i].octave;
        const float radius = th * CurrentFrame.mvScaleFactors[nLastOctave];
        std::vector<size_t> vIndices2;
        if (bForward) vIndices2 = G.area(u, v, radius, nLastOctave);
        else if (bBackward) vIndices2 = G.area(u, v, radius, 0, nLastOctave);
        else vIndices2 = G.area(u, v, radius, nLastOctave - 1, nLastOctave + 1);
        if (vIndices2.empty()) continue;
        const auto dMP = pMP->GetDescriptor();
        int bestDist = 256, bestIdx2 = -1;
        for (size_t k = 0; k < vIndices2.size(); k++) {
            const size_t i2 = vIndices2[k];
            if (CurrentFrame.mvpMapPoints[i2])
                if (CurrentFrame.mvpMapPoints[i2]->Observations() > 0) continue;
            if (CurrentFrame.mvuRight[i2] > 0) {
                const float ur = u - CurrentFrame.mbf * invzc;
                const float er = fabsf(ur - CurrentFrame.mvuRight[i2]);
                if (er > radius) continue;
            }
            const int dist = ham(dMP, CurrentFrame.mDescriptors.ptr((int)i2));
            if (dist < bestDist) { bestDist = dist; bestIdx2 = (int)i2; }
        }
        if (bestDist <= ORC_TH_HIGH) {
            CurrentFrame.mvpMapPoints[bestIdx2] = pMP;
            nmatches++;
            if (checkOri) rotHist[rot_bin(LastFrame.mvKeysUn[i].angle, CurrentFrame.mvKeysUn[bestIdx2].angle)].push_back(bestIdx2);
        }
    }
    if (checkOri) {
        int ind1, ind2, ind3;
        three_maxima(rotHist, ind1, ind2, ind3);
        for (int i = 0; i < ORC_HISTO_LENGTH; i++)
            if (i != ind1 && i != ind2 && i != ind3)
                for (size_t j = 0; j < rotHist[i].size(); j++) { CurrentFrame.mvpMapPoints[rotHist[i][j]] = nullptr; nmatches--; }
    }
    return nmatches;
}

// :1523-1650
template <class FrameT, class KF, class MP>
int SearchByProjection(FrameT &CurrentFrame, KF *pKF, const std::set<MP *> &sAlreadyFound, const float th, const int ORBdist, bool checkOri)
{
    Grid G; frame_grid(CurrentFrame, G);
    int nmatches = 0;
    const Rt cw = rt_of_4x4(CurrentFrame.mTcw);
    const P3 Ow = rt_centre(cw);
    std::vector<int> rotHist[ORC_HISTO_LENGTH];
    const std::vector<MP *> vpMPs = pKF->GetMapPointMatches();
    for (size_t i = 0; i < vpMPs.size(); i++) {
        MP *pMP = vpMPs[i];
        if (!pMP) continue;
        if (pMP->isBad() || sAlreadyFound.count(pMP)) continue;
        const P3 x3Dw = col3(pMP->GetWorldPos());
        const P3 x3Dc = rt_apply(cw, x3Dw);
        const float xc = x3Dc.v[0], yc = x3Dc.v[1];
        const float invzc = 1.0 / x3Dc.v[2];
        const float u = CurrentFrame.fx * xc * invzc + CurrentFrame.cx;
        const float v = CurrentFrame.fy * yc * invzc + CurrentFrame.cy;
        if (u < CurrentFrame.mnMinX || u > CurrentFrame.mnMaxX) continue;
        if (v < CurrentFrame.mnMinY || v > CurrentFrame.mnMaxY) continue;
        const P3 PO = minus(x3Dw, Ow);
        const float dist3D = l2norm(PO);
        const float maxDistance = pMP->GetMaxDistanceInvariance();
        const float minDistance = pMP->GetMinDistanceInvariance();
        if (dist3D < minDistance || dist3D > maxDistance) continue;
        const int nPredictedLevel = pMP->PredictScale(dist3D, &CurrentFrame);
        const float radius = th * CurrentFrame.mvScaleFactors[nPredictedLevel];
        const std::vector<size_t> vIndices2 = G.area(u, v, radius, nPredictedLevel - 1, nPredictedLevel + 1);
        if (vIndices2.empty()) continue;
        const auto dMP = pMP->GetDescriptor();
        int bestDist = 256, bestIdx2 = -1;
        for (size_t k = 0; k < vIndices2.size(); k++) {
            const size_t i2 = vIndices2[k];
            if (CurrentFrame.mvpMapPoints[i2]) continue;
            const int dist = ham(dMP, CurrentFrame.mDescriptors.ptr((int)i2));
            if (dist < bestDist) { bestDist = dist; bestIdx2 = (int)i2; }
        }
        if (bestIdx2 >= 0 && bestDist <= ORBdist) {
            CurrentFrame.mvpMapPoints[bestIdx2] = pMP;
            nmatches++;
            if (checkOri) rotHist[rot_bin(pKF->mvKeysUn[i].angle, CurrentFrame.mvKeysUn[bestIdx2].angle)].push_back(bestIdx2);
        }
    }
    if (checkOri) {
        int ind1, ind2, ind3;
        three_maxima(rotHist, ind1, ind2, ind3);
        for (int i = 0; i < ORC_HISTO_LENGTH; i++)
            if (i != ind1 && i != ind2 && i != ind3)
                for (size_t j = 0; j < rotHist[i].size(); j++) { CurrentFrame.mvpMapPoints[rotHist[i][j]] = nullptr; nmatches--; }
    }
    return nmatches;
}

// :290-403
template <class KF, class M, class MP>
int SearchByProjection(KF *pKF, const M &Scw, const std::vector<MP *> &vpPoints, std::vector<MP *> &vpMatched, int th)
{
    Grid G; keyframe_grid(*pKF, G);
    const float fx = pKF->fx, fy = pKF->fy, cx = pKF->cx, cy = pKF->cy;
    const Rt cw = rt_of_sim3(Scw);
    const P3 Ow = rt_centre(cw);
    std::set<MP *> spAlreadyFound(vpMatched.begin(), vpMatched.end());
    spAlreadyFound.erase(static_cast<MP *>(nullptr));
    int nmatches = 0;
    for (int iMP = 0, iendMP = (int)vpPoints.size(); iMP < iendMP; iMP++) {
        MP *pMP = vpPoints[iMP];
        if (pMP->isBad() || spAlreadyFound.count(pMP)) continue;
        const P3 p3Dw = col3(pMP->GetWorldPos());
        const P3 p3Dc = rt_apply(cw, p3Dw);
        if (p3Dc.v[2] < 0.0) continue;
        const float invz = 1 / p3Dc.v[2];
        const float x = p3Dc.v[0] * invz;
        const float y = p3Dc.v[1] * invz;
        const float u = fx * x + cx;
        const float v = fy * y + cy;
        if (!pKF->IsInImage(u, v)) continue;
        const float maxDistance = pMP->GetMaxDistanceInvariance();
        const float minDistance = pMP->GetMinDistanceInvariance();
        const P3 PO = minus(p3Dw, Ow);
        const float dist = l2norm(PO);
        if (dist < minDistance || dist > maxDistance) continue;
        const P3 Pn = col3(pMP->GetNormal());
        if (ddot(PO, Pn) < 0.5 * dist) continue;
        const int nPredictedLevel = pMP->PredictScale(dist, pKF);
        const float radius = th * pKF->mvScaleFactors[nPredictedLevel];
        const std::vector<size_t> vIndices = G.area(u, v, radius);
        if (vIndices.empty()) continue;
        const auto dMP = pMP->GetDescriptor();
        int bestDist = 256, bestIdx = -1;
        for (size_t k = 0; k < vIndices.size(); k++) {
            const size_t idx = vIndices[k];
            if (vpMatched[idx]) continue;
            const int kpLevel = pKF->mvKeysUn[idx].octave;
            if (kpLevel < nPredictedLevel - 1 || kpLevel > nPredictedLevel) continue;
            const int d = ham(dMP, pKF->mDescriptors.ptr((int)idx));
            if (d < bestDist) { bestDist = d; bestIdx = (int)idx; }
        }
        if (bestDist <= ORC_TH_LOW) { vpMatched[bestIdx] = pMP; nmatches++; }
    }
    return nmatches;
}

// :159-288
template <class KF, class FrameT, class MP>
int SearchByBoW(KF *pKF, FrameT &F, std::vector<MP *> &vpMapPointMatches, float nnratio, bool checkOri)
{
    const std::vector<MP *> vpMapPointsKF = pKF->GetMapPointMatches();
    vpMapPointMatches = std::vector<MP *>(F.N, static_cast<MP *>(nullptr));
    int nmatches = 0;
    std::vector<int> rotHist[ORC_HISTO_LENGTH];
    auto KFit = pKF->mFeatVec.begin(), KFend = pKF->mFeatVec.end();
    auto Fit = F.mFeatVec.begin(), Fend = F.mFeatVec.end();
    while (KFit != KFend && Fit != Fend) {
        if (KFit->first == Fit->first) {
            const std::vector<unsigned int> vIndicesKF = KFit->second, vIndicesF = Fit->second;
            for (size_t iKF = 0; iKF < vIndicesKF.size(); iKF++) {
                const unsigned int realIdxKF = vIndicesKF[iKF];
                MP *pMP = vpMapPointsKF[realIdxKF];
                if (!pMP) continue;
                if (pMP->isBad()) continue;
                const uint8_t *dKF = pKF->mDescriptors.ptr((int)realIdxKF);
                int bestDist1 = 256, bestIdxF = -1, bestDist2 = 256;
                for (size_t iF = 0; iF < vIndicesF.size(); iF++) {
                    const unsigned int realIdxF = vIndicesF[iF];
                    if (vpMapPointMatches[realIdxF]) continue;
                    const int dist = orc_hamming256(dKF, F.mDescriptors.ptr((int)realIdxF));
                    if (dist < bestDist1) { bestDist2 = bestDist1; bestDist1 = dist; bestIdxF = (int)realIdxF; }
                    else if (dist < bestDist2) bestDist2 = dist;
                }
                if (bestDist1 <= ORC_TH_LOW) {
                    if ((float)bestDist1 < nnratio * (float)bestDist2) {
                        vpMapPointMatches[bestIdxF] = pMP;
                        if (checkOri) rotHist[rot_bin(pKF->mvKeysUn[realIdxKF].angle, F.mvKeys[bestIdxF].angle)].push_back(bestIdxF);
                        nmatches++;
                    }
                }
            }
            KFit++; Fit++;
        } else if (KFit->first < Fit->first) {
            KFit = pKF->mFeatVec.lower_bound(Fit->first);
        } else {
            Fit = F.mFeatVec.lower_bound(KFit->first);
        }
    }
    if (checkOri) {
        int ind1, ind2, ind3;
        three_maxima(rotHist, ind1, ind2, ind3);
        for (int i = 0; i < ORC_HISTO_LENGTH; i++) {
            if (i == ind1 || i == ind2 || i == ind3) continue;
            for (size_t j = 0; j < rotHist[i].size(); j++) { vpMapPointMatches[rotHist[i][j]] = nullptr; nmatches--; }
        }
    }
    return nmatches;
}

// :573-706
template <class KF, class MP>
int SearchByBoW(KF *pKF1, KF *pKF2, std::vector<MP *> &vpMatches12, float nnratio, bool checkOri)
{
    const std::vector<MP *> vpMapPoints1 = pKF1->GetMapPointMatches(), vpMapPoints2 = pKF2->GetMapPointMatches();
    vpMatches12 = std::vector<MP *>(vpMapPoints1.size(), static_cast<MP *>(nullptr));
    std::vector<bool> vbMatched2(vpMapPoints2.size(), false);
    std::vector<int> rotHist[ORC_HISTO_LENGTH];
    int nmatches = 0;
    auto f1it = pKF1->mFeatVec.begin(), f1end = pKF1->mFeatVec.end();
    auto f2it = pKF2->mFeatVec.begin(), f2end = pKF2->mFeatVec.end();
    while (f1it != f1end && f2it != f2end) {
        if (f1it->first == f2it->first) {
            for (size_t i1 = 0; i1 < f1it->second.size(); i1++) {
                const size_t idx1 = f1it->second[i1];
                MP *pMP1 = vpMapPoints1[idx1];
                if (!pMP1) continue;
                if (pMP1->isBad()) continue;
                const uint8_t *d1 = pKF1->mDescriptors.ptr((int)idx1);
                int bestDist1 = 256, bestIdx2 = -1, bestDist2 = 256;
                for (size_t i2 = 0; i2 < f2it->second.size(); i2++) {
                    const size_t idx2 = f2it->second[i2];
                    MP *pMP2 = vpMapPoints2[idx2];
                    if (vbMatched2[idx2] || !pMP2) continue;
                    if (pMP2->isBad()) continue;
                    const int dist = orc_hamming256(d1, pKF2->mDescriptors.ptr((int)idx2));
                    if (dist < bestDist1) { bestDist2 = bestDist1; bestDist1 = dist; bestIdx2 = (int)idx2; }
                    else if (dist < bestDist2) bestDist2 = dist;
                }
                if (bestDist1 < ORC_TH_LOW) {
                    if ((float)bestDist1 < nnratio * (float)bestDist2) {
                        vpMatches12[idx1] = vpMapPoints2[bestIdx2];
                        vbMatched2[bestIdx2] = true;
                        if (checkOri) rotHist[rot_bin(pKF1->mvKeysUn[idx1].angle, pKF2->mvKeysUn[bestIdx2].angle)].push_back((int)idx1);
                        nmatches++;
                    }
                }
            }
            f1it++; f2it++;
        } else if (f1it->first < f2it->first) {
            f1it = pKF1->mFeatVec.lower_bound(f2it->first);
        } else {
            f2it = pKF2->mFeatVec.lower_bound(f1it->first);
        }
    }
    if (checkOri) {
        int ind1, ind2, ind3;
        three_maxima(rotHist, ind1, ind2, ind3);
        for (int i = 0; i < ORC_HISTO_LENGTH; i++) {
            if (i == ind1 || i == ind2 || i == ind3) continue;
            for (size_t j = 0; j < rotHist[i].size(); j++) { vpMatches12[rotHist[i][j]] = nullptr; nmatches--; }
        }
    }
    return nmatches;
}

// :422-571
template <class FrameT, class PT>
int SearchForInitialization(FrameT &F1, FrameT &F2, std::vector<PT> &vbPrevMatched, std::vector<int> &vnMatches12, int windowSize,
                            float nnratio, bool checkOri)
{
    Grid G; frame_grid(F2, G);
    int nmatches = 0;
    vnMatches12 = std::vector<int>(F1.mvKeysUn.size(), -1);
    std::vector<int> rotHist[ORC_HISTO_LENGTH];
    std::vector<int> vMatchedDistance(F2.mvKeysUn.size(), INT_MAX), vnMatches21(F2.mvKeysUn.size(), -1);
    for (size_t i1 = 0; i1 < F1.mvKeysUn.size(); i1++) {
        const int level1 = F1.mvKeysUn[i1].octave;
        if (level1 > 0) continue;
        const std::vector<size_t> vIndices2 = G.area(vbPrevMatched[i1].x, vbPrevMatched[i1].y, (float)windowSize, level1, level1);
        if (vIndices2.empty()) continue;
        const uint8_t *d1 = F1.mDescriptors.ptr((int)i1);
        int bestDist = INT_MAX, bestDist2 = INT_MAX, bestIdx2 = -1;
        for (size_t k = 0; k < vIndices2.size(); k++) {
            const size_t i2 = vIndices2[k];
            const int dist = orc_hamming256(d1, F2.mDescriptors.ptr((int)i2));
            if (vMatchedDistance[i2] <= dist) continue;
            if (dist < bestDist) { bestDist2 = bestDist; bestDist = dist; bestIdx2 = (int)i2; }
            else if (dist < bestDist2) bestDist2 = dist;
        }
        if (bestDist <= ORC_TH_LOW) {
            if (bestDist < (float)bestDist2 * nnratio) {
                if (vnMatches21[bestIdx2] >= 0) { vnMatches12[vnMatches21[bestIdx2]] = -1; nmatches--; }
                vnMatches12[i1] = bestIdx2;
                vnMatches21[bestIdx2] = (int)i1;
                vMatchedDistance[bestIdx2] = bestDist;
                nmatches++;
                if (checkOri) rotHist[rot_bin(F1.mvKeysUn[i1].angle, F2.mvKeysUn[bestIdx2].angle)].push_back((int)i1);
            }
        }
    }
    if (checkOri) {
        int ind1, ind2, ind3;
        three_maxima(rotHist, ind1, ind2, ind3);
        for (int i = 0; i < ORC_HISTO_LENGTH; i++) {
            if (i == ind1 || i == ind2 || i == ind3) continue;
            for (size_t j = 0; j < rotHist[i].size(); j++) {
                const int idx1 = rotHist[i][j];
                if (vnMatches12[idx1] >= 0) { vnMatches12[idx1] = -1; nmatches--; }
            }
        }
    }
    for (size_t i1 = 0; i1 < vnMatches12.size(); i1++)
        if (vnMatches12[i1] >= 0) vbPrevMatched[i1] = F2.mvKeysUn[vnMatches12[i1]].pt;
    return nmatches;
}

// :140-157
template <class KP, class M, class KF>
inline bool CheckDistEpipolarLine(const KP &kp1, const KP &kp2, const M &F12, const KF *pKF2)
{
    const float a = kp1.pt.x * F12.template at<float>(0, 0) + kp1.pt.y * F12.template at<float>(1, 0) + F12.template at<float>(2, 0);
    const float b = kp1.pt.x * F12.template at<float>(0, 1) + kp1.pt.y * F12.template at<float>(1, 1) + F12.template at<float>(2, 1);
    const float c = kp1.pt.x * F12.template at<float>(0, 2) + kp1.pt.y * F12.template at<float>(1, 2) + F12.template at<float>(2, 2);
    const float num = a * kp2.pt.x + b * kp2.pt.y + c;
    const float den = a * a + b * b;
    if (den == 0) return false;
    const float dsqr = num * num / den;
    return dsqr < 3.84 * pKF2->mvLevelSigma2[kp2.octave];
}

// :708-874
template <class KF, class M>
int SearchForTriangulation(KF *pKF1, KF *pKF2, const M &F12, std::vector<std::pair<size_t, size_t> > &vMatchedPairs, const bool bOnlyStereo,
                           bool checkOri)
{
    const P3 Cw = col3(pKF1->GetCameraCenter());
    const Rt w2 = rt_of(pKF2->GetRotation(), pKF2->GetTranslation());
    const P3 C2 = rt_apply(w2, Cw);
    const float invz = 1.0f / C2.v[2];
    const float ex = pKF2->fx * C2.v[0] * invz + pKF2->cx;
    const float ey = pKF2->fy * C2.v[1] * invz + pKF2->cy;
    int nmatches = 0;
    std::vector<int> vMatches12(pKF1->N, -1);
    std::vector<int> rotHist[ORC_HISTO_LENGTH];
    auto f1it = pKF1->mFeatVec.begin(), f1end = pKF1->mFeatVec.end();
    auto f2it = pKF2->mFeatVec.begin(), f2end = pKF2->mFeatVec.end();
    while (f1it != f1end && f2it != f2end) {
        if (f1it->first == f2it->first) {
            for (size_t i1 = 0; i1 < f1it->second.size(); i1++) {
                const size_t idx1 = f1it->second[i1];
                if (pKF1->GetMapPoint(idx1)) continue;
                const bool bStereo1 = pKF1->mvuRight[idx1] >= 0;
                if (bOnlyStereo && !bStereo1) continue;
                const auto &kp1 = pKF1->mvKeysUn[idx1];
                const uint8_t *d1 = pKF1->mDescriptors.ptr((int)idx1);
                int bestDist = ORC_TH_LOW, bestIdx2 = -1;
                for (size_t i2 = 0; i2 < f2it->second.size(); i2++) {
                    const size_t idx2 = f2it->second[i2];
                    if (pKF2->GetMapPoint(idx2)) continue;  // vbMatched2 is never set in the reference (:728, :776)
                    const bool bStereo2 = pKF2->mvuRight[idx2] >= 0;
                    if (bOnlyStereo && !bStereo2) continue;
                    const int dist = orc_hamming256(d1, pKF2->mDescriptors.ptr((int)idx2));
                    if (dist > ORC_TH_LOW || dist > bestDist) continue;
                    const auto &kp2 = pKF2->mvKeysUn[idx2];
                    if (!bStereo1 && !bStereo2) {
                        const float distex = ex - kp2.pt.x, distey = ey - kp2.pt.y;
                        if (distex * distex + distey * distey < 100 * pKF2->mvScaleFactors[kp2.octave]) continue;
                    }
                    if (CheckDistEpipolarLine(kp1, kp2, F12, pKF2)) { bestIdx2 = (int)idx2; bestDist = dist; }
                }
                if (bestIdx2 >= 0) {
                    vMatches12[idx1] = bestIdx2;
                    nmatches++;
                    if (checkOri) rotHist[rot_bin(kp1.angle, pKF2->mvKeysUn[bestIdx2].angle)].push_back((int)idx1);
                }
            }
            f1it++; f2it++;
        } else if (f1it->first < f2it->first) {
            f1it = pKF1->mFeatVec.lower_bound(f2it->first);
        } else {
            f2it = pKF2->mFeatVec.lower_bound(f1it->first);
        }
    }
    if (checkOri) {
        int ind1, ind2, ind3;
        three_maxima(rotHist, ind1, ind2, ind3);
        for (int i = 0; i < ORC_HISTO_LENGTH; i++) {
            if (i == ind1 || i == ind2 || i == ind3) continue;
            for (size_t j = 0; j < rotHist[i].size(); j++) { vMatches12[rotHist[i][j]] = -1; nmatches--; }
        }
    }
    vMatchedPairs.clear();
    for (size_t i = 0; i < vMatches12.size(); i++)
        if (vMatches12[i] >= 0) vMatchedPairs.push_back(std::make_pair(i, (size_t)vMatches12[i]));
    return nmatches;
}

// :876-1026
template <class KF, class MP>
int Fuse(KF *pKF, const std::vector<MP *> &vpMapPoints, const float th)
{
    Grid G; keyframe_grid(*pKF, G);
    const Rt cw = rt_of(pKF->GetRotation(), pKF->GetTranslation());
    const float fx = pKF->fx, fy = pKF->fy, cx = pKF->cx, cy = pKF->cy, bf = pKF->mbf;
    const P3 Ow = col3(pKF->GetCameraCenter());
    int nFused = 0;
    for (int i = 0, n = (int)vpMapPoints.size(); i < n; i++) {
        MP *pMP = vpMapPoints[i];
        if (!pMP) continue;
        if (pMP->isBad() || pMP->IsInKeyFrame(pKF)) continue;
        const P3 p3Dw = col3(pMP->GetWorldPos());
        const P3 p3Dc = rt_apply(cw, p3Dw);
        if (p3Dc.v[2] < 0.0f) continue;
        const float invz = 1 / p3Dc.v[2];
        const float x = p3Dc.v[0] * invz, y = p3Dc.v[1] * invz;
        const float u = fx * x + cx, v = fy * y + cy;
        if (!pKF->IsInImage(u, v)) continue;
        const float ur = u - bf * invz;
        const float maxDistance = pMP->GetMaxDistanceInvariance(), minDistance = pMP->GetMinDistanceInvariance();
        const P3 PO = minus(p3Dw, Ow);
        const float dist3D = l2norm(PO);
        if (dist3D < minDistance || dist3D > maxDistance) continue;
        if (ddot(PO, col3(pMP->GetNormal())) < 0.5 * dist3D) continue;
        const int nPredictedLevel = pMP->PredictScale(dist3D, pKF);
        const float radius = th * pKF->mvScaleFactors[nPredictedLevel];
        const std::vector<size_t> vIndices = G.area(u, v, radius);
        if (vIndices.empty()) continue;
        const auto dMP = pMP->GetDescriptor();
        int bestDist = 256, bestIdx = -1;
        for (size_t k = 0; k < vIndices.size(); k++) {
            const size_t idx = vIndices[k];
            const auto &kp = pKF->mvKeysUn[idx];
            const int kpLevel = kp.octave;
            if (kpLevel < nPredictedLevel - 1 || kpLevel > nPredictedLevel) continue;
            if (pKF->mvuRight[idx] >= 0) {
                const float ex = u - kp.pt.x, ey = v - kp.pt.y, er = ur - pKF->mvuRight[idx];
                const float e2 = ex * ex + ey * ey + er * er;
                if (e2 * pKF->mvInvLevelSigma2[kpLevel] > 7.8) continue;
            } else {
                const float ex = u - kp.pt.x, ey = v - kp.pt.y;
                const float e2 = ex * ex + ey * ey;
                if (e2 * pKF->mvInvLevelSigma2[kpLevel] > 5.99) continue;
            }
            const int dist = ham(dMP, pKF->mDescriptors.ptr((int)idx));
            if (dist < bestDist) { bestDist = dist; bestIdx = (int)idx; }
        }
        if (bestDist <= ORC_TH_LOW) {
            MP *pMPinKF = pKF->GetMapPoint(bestIdx);
            if (pMPinKF) {
                if (!pMPinKF->isBad()) {
                    if (pMPinKF->Observations() > pMP->Observations()) pMP->Replace(pMPinKF);
                    else pMPinKF->Replace(pMP);
                }
            } else {
                pMP->AddObservation(pKF, bestIdx);
                pKF->AddMapPoint(pMP, bestIdx);
            }
            nFused++;
        }
    }
    return nFused;
}

// :1028-1151
template <class KF, class M, class MP>
int Fuse(KF *pKF, const M &Scw, const std::vector<MP *> &vpPoints, float th, std::vector<MP *> &vpReplacePoint)
{
    Grid G; keyframe_grid(*pKF, G);
    const float fx = pKF->fx, fy = pKF->fy, cx = pKF->cx, cy = pKF->cy;
    const Rt cw = rt_of_sim3(Scw);
    const P3 Ow = rt_centre(cw);
    const std::set<MP *> spAlreadyFound = pKF->GetMapPoints();
    int nFused = 0;
    for (int iMP = 0, n = (int)vpPoints.size(); iMP < n; iMP++) {
        MP *pMP = vpPoints[iMP];
        if (pMP->isBad() || spAlreadyFound.count(pMP)) continue;
        const P3 p3Dw = col3(pMP->GetWorldPos());
        const P3 p3Dc = rt_apply(cw, p3Dw);
        if (p3Dc.v[2] < 0.0f) continue;
        const float invz = 1.0 / p3Dc.v[2];
        const float x = p3Dc.v[0] * invz, y = p3Dc.v[1] * invz;
        const float u = fx * x + cx, v = fy * y + cy;
        if (!pKF->IsInImage(u, v)) continue;
        const float maxDistance = pMP->GetMaxDistanceInvariance(), minDistance = pMP->GetMinDistanceInvariance();
        const P3 PO = minus(p3Dw, Ow);
        const float dist3D = l2norm(PO);
        if (dist3D < minDistance || dist3D > maxDistance) continue;
        if (ddot(PO, col3(pMP->GetNormal())) < 0.5 * dist3D) continue;
        const int nPredictedLevel = pMP->PredictScale(dist3D, pKF);
        const float radius = th * pKF->mvScaleFactors[nPredictedLevel];
        const std::vector<size_t> vIndices = G.area(u, v, radius);
        if (vIndices.empty()) continue;
        const auto dMP = pMP->GetDescriptor();
        int bestDist = INT_MAX, bestIdx = -1;
        for (size_t k = 0; k < vIndices.size(); k++) {
            const size_t idx = vIndices[k];
            const int kpLevel = pKF->mvKeysUn[idx].octave;
            if (kpLevel < nPredictedLevel - 1 || kpLevel > nPredictedLevel) continue;
            const int dist = ham(dMP, pKF->mDescriptors.ptr((int)idx));
            if (dist < bestDist) { bestDist = dist; bestIdx = (int)idx; }
        }
        if (bestDist <= ORC_TH_LOW) {
            MP *pMPinKF = pKF->GetMapPoint(bestIdx);
            if (pMPinKF) {
                if (!pMPinKF->isBad()) vpReplacePoint[iMP] = pMPinKF;
            } else {
                pMP->AddObservation(pKF, bestIdx);
                pKF->AddMapPoint(pMP, bestIdx);
            }
            nFused++;
        }
    }
    return nFused;
}

// :1153-1377
template <class KF, class MP, class M>
int SearchBySim3(KF *pKF1, KF *pKF2, std::vector<MP *> &vpMatches12, const float &s12, const M &R12, const M &t12, const float th)
{
    const float fx = pKF1->fx, fy = pKF1->fy, cx = pKF1->cx, cy = pKF1->cy;
    const Rt w1 = rt_of(pKF1->GetRotation(), pKF1->GetTranslation());
    const Rt w2 = rt_of(pKF2->GetRotation(), pKF2->GetTranslation());
    Rt s12m, s21m;  // [sR12 | t12] and [sR21 | t21]
    const float is = (float)(1.0 / (double)s12);
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) {
            s12m.R[i][j] = R12.template at<float>(i, j) * s12;
            s21m.R[i][j] = R12.template at<float>(j, i) * is;
        }
    for (int i = 0; i < 3; i++) s12m.t[i] = t12.template at<float>(i, 0);
    for (int i = 0; i < 3; i++) {
        float acc = s21m.R[i][0] * s12m.t[0];
        acc = acc + s21m.R[i][1] * s12m.t[1];
        acc = acc + s21m.R[i][2] * s12m.t[2];
        s21m.t[i] = -acc;
    }
    const std::vector<MP *> vpMapPoints1 = pKF1->GetMapPointMatches(), vpMapPoints2 = pKF2->GetMapPointMatches();
    const int N1 = (int)vpMapPoints1.size(), N2 = (int)vpMapPoints2.size();
    std::vector<bool> vbAlreadyMatched1(N1, false), vbAlreadyMatched2(N2, false);
    for (int i = 0; i < N1; i++) {
        MP *pMP = vpMatches12[i];
        if (pMP) {
            vbAlreadyMatched1[i] = true;
            const int idx2 = pMP->GetIndexInKeyFrame(pKF2);
            if (idx2 >= 0 && idx2 < N2) vbAlreadyMatched2[idx2] = true;
        }
    }
    std::vector<int> vnMatch1(N1, -1), vnMatch2(N2, -1);
    Grid G2; keyframe_grid(*pKF2, G2);
    for (int i1 = 0; i1 < N1; i1++) {
        MP *pMP = vpMapPoints1[i1];
        if (!pMP || vbAlreadyMatched1[i1]) continue;
        if (pMP->isBad()) continue;
        const P3 p3Dc1 = rt_apply(w1, col3(pMP->GetWorldPos()));
        const P3 p3Dc2 = rt_apply(s21m, p3Dc1);
        if (p3Dc2.v[2] < 0.0) continue;
        const float invz = 1.0 / p3Dc2.v[2];
        const float x = p3Dc2.v[0] * invz, y = p3Dc2.v[1] * invz;
        const float u = fx * x + cx, v = fy * y + cy;
        if (!pKF2->IsInImage(u, v)) continue;
        const float maxDistance = pMP->GetMaxDistanceInvariance(), minDistance = pMP->GetMinDistanceInvariance();
        const float dist3D = l2norm(p3Dc2);
        if (dist3D < minDistance || dist3D > maxDistance) continue;
        const int nPredictedLevel = pMP->PredictScale(dist3D, pKF2);
        const float radius = th * pKF2->mvScaleFactors[nPredictedLevel];
        const std::vector<size_t> vIndices = G2.area(u, v, radius);
        if (vIndices.empty()) continue;
        const auto dMP = pMP->GetDescriptor();
        int bestDist = INT_MAX, bestIdx = -1;
        for (size_t k = 0; k < vIndices.size(); k++) {
            const size_t idx = vIndices[k];
            const auto &kp = pKF2->mvKeysUn[idx];
            if (kp.octave < nPredictedLevel - 1 || kp.octave > nPredictedLevel) continue;
            const int dist = ham(dMP, pKF2->mDescriptors.ptr((int)idx));
            if (dist < bestDist) { bestDist = dist; bestIdx = (int)idx; }
        }
        if (bestDist <= ORC_TH_HIGH) vnMatch1[i1] = bestIdx;
    }
    Grid G1; keyframe_grid(*pKF1, G1);
    for (int i2 = 0; i2 < N2; i2++) {
        MP *pMP = vpMapPoints2[i2];
        if (!pMP || vbAlreadyMatched2[i2]) continue;
        if (pMP->isBad()) continue;
        const P3 p3Dc2 = rt_apply(w2, col3(pMP->GetWorldPos()));
        const P3 p3Dc1 = rt_apply(s12m, p3Dc2);
        if (p3Dc1.v[2] < 0.0) continue;
        const float invz = 1.0 / p3Dc1.v[2];
        const float x = p3Dc1.v[0] * invz, y = p3Dc1.v[1] * invz;
        const float u = fx * x + cx, v = fy * y + cy;
        if (!pKF1->IsInImage(u, v)) continue;
        const float maxDistance = pMP->GetMaxDistanceInvariance(), minDistance = pMP->GetMinDistanceInvariance();
        const float dist3D = l2norm(p3Dc1);
        if (dist3D < minDistance || dist3D > maxDistance) continue;
        const int nPredictedLevel = pMP->PredictScale(dist3D, pKF1);
        const float radius = th * pKF1->mvScaleFactors[nPredictedLevel];
        const std::vector<size_t> vIndices = G1.area(u, v, radius);
        if (vIndices.empty()) continue;
        const auto dMP = pMP->GetDescriptor();
        int bestDist = INT_MAX, bestIdx = -1;
        for (size_t k = 0; k < vIndices.size(); k++) {
            const size_t idx = vIndices[k];
            const auto &kp = pKF1->mvKeysUn[idx];
            if (kp.octave < nPredictedLevel - 1 || kp.octave > nPredictedLevel) continue;
            const int dist = ham(dMP, pKF1->mDescriptors.ptr((int)idx));
            if (dist < bestDist) { bestDist = dist; bestIdx = (int)idx; }
        }
        if (bestDist <= ORC_TH_HIGH) vnMatch2[i2] = bestIdx;
    }
    int nFound = 0;
    for (int i1 = 0; i1 < N1; i1++) {
        const int idx2 = vnMatch1[i1];
        if (idx2 >= 0) {
            const int idx1 = vnMatch2[idx2];
            if (idx1 == i1) { vpMatches12[i1] = vpMapPoints2[idx2]; nFound++; }
        }
    }
    return nFound;
}

}  // namespace orc_frames

#endif
