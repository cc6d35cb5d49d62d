// oracle/locref_loam.hpp
//
// TEST INFRASTRUCTURE ONLY — CPU restatement of LoamFeatureExtract::Extract / ExtractFromSector
// (LocUtils/src/model/feature_extract/loam_feature_extract.cpp:19-151), the per-ring curvature feature picker that feeds
// LoamRegistration. Nothing under loc_lib_amd/ or include/ uses this file.
// PARITY UNPINNED: the reference has no test or golden vector for it. Quirks kept on purpose:
//   * every sector drops its last element (the sub-vector's end iterator is `begin + sector_end`, :83-84);
//   * the 21st pick of a sector is marked as picked but emitted neither as an edge nor as a surface point (:110-118);
//   * neighbour marking walks ring indices beyond the sector (:120-138);
//   * curvature differences are float32 sums evaluated left to right, widened to double only on assignment (:50-67);
//   * rings with fewer than 131 points are skipped (:40-43).
// The one thing the reference leaves open is the order of EQUAL curvatures (std::sort is unstable): `order = 0` uses this
// toolchain's std::sort on the value alone, like the reference; `order = 1` breaks ties by ascending id (what the GPU does).
#pragma once
#include <algorithm>
#include <cstdint>
#include <vector>

#include "locref_filters.hpp"

namespace locref {

struct IdAndValue {
    int id_ = 0;
    double value_ = 0;
};

static inline void LoamExtractFromSector(const std::vector<PointXYZI>& pc_in, std::vector<IdAndValue>& cloud_curvature, int order,
                                         std::vector<PointXYZI>& edge, std::vector<PointXYZI>& surf) {
    if (order == 0)
        std::sort(cloud_curvature.begin(), cloud_curvature.end(), [](const IdAndValue& a, const IdAndValue& b) { return a.value_ < b.value_; });
    else
        std::sort(cloud_curvature.begin(), cloud_curvature.end(),
                  [](const IdAndValue& a, const IdAndValue& b) { return a.value_ < b.value_ || (a.value_ == b.value_ && a.id_ < b.id_); });
    int largest_picked_num = 0;
    std::vector<int> picked_points;
    for (int i = (int)cloud_curvature.size() - 1; i >= 0; --i) {
        const int ind = cloud_curvature[i].id_;
        if (std::find(picked_points.begin(), picked_points.end(), ind) == picked_points.end()) {
            if (cloud_curvature[i].value_ <= 0.1) break;
            largest_picked_num++;
            picked_points.push_back(ind);
            if (largest_picked_num <= 20) edge.push_back(pc_in[ind]);
            else break;
            for (int k = 1; k <= 5; k++) {
                const double diffX = pc_in[ind + k].x - pc_in[ind + k - 1].x;  // float difference, widened
                const double diffY = pc_in[ind + k].y - pc_in[ind + k - 1].y;
                const double diffZ = pc_in[ind + k].z - pc_in[ind + k - 1].z;
                if (diffX * diffX + diffY * diffY + diffZ * diffZ > 0.05) break;
                picked_points.push_back(ind + k);
            }
            for (int k = -1; k >= -5; k--) {
                const double diffX = pc_in[ind + k].x - pc_in[ind + k + 1].x;
                const double diffY = pc_in[ind + k].y - pc_in[ind + k + 1].y;
                const double diffZ = pc_in[ind + k].z - pc_in[ind + k + 1].z;
                if (diffX * diffX + diffY * diffY + diffZ * diffZ > 0.05) break;
                picked_points.push_back(ind + k);
            }
        }
    }
    for (int i = 0; i <= (int)cloud_curvature.size() - 1; i++) {
        const int ind = cloud_curvature[i].id_;
        if (std::find(picked_points.begin(), picked_points.end(), ind) == picked_points.end()) surf.push_back(pc_in[ind]);
    }
}

// pts: x, y, z, intensity (already converted from the uint8 field); ring[i] in [0, num_scan).
static inline void LoamExtract(const PointXYZI* pts, const uint8_t* ring, size_t n, int num_scan, int order, std::vector<PointXYZI>& edge,
                               std::vector<PointXYZI>& surf) {
    std::vector<std::vector<PointXYZI>> lines(num_scan);
    for (size_t i = 0; i < n; ++i)
        if ((int)ring[i] < num_scan) lines[ring[i]].push_back(pts[i]);  // (the reference indexes unchecked; out-of-range rings would be UB)
    for (int i = 0; i < num_scan; ++i) {
        const std::vector<PointXYZI>& L = lines[i];
        if (L.size() < 131) continue;
        std::vector<IdAndValue> cloud_curvature;
        const int total_points = (int)L.size() - 10;
        for (int j = 5; j < (int)L.size() - 5; j++) {
            const float fx = L[j - 5].x + L[j - 4].x + L[j - 3].x + L[j - 2].x + L[j - 1].x - 10 * L[j].x + L[j + 1].x + L[j + 2].x + L[j + 3].x +
                             L[j + 4].x + L[j + 5].x;
            const float fy = L[j - 5].y + L[j - 4].y + L[j - 3].y + L[j - 2].y + L[j - 1].y - 10 * L[j].y + L[j + 1].y + L[j + 2].y + L[j + 3].y +
                             L[j + 4].y + L[j + 5].y;
            const float fz = L[j - 5].z + L[j - 4].z + L[j - 3].z + L[j - 2].z + L[j - 1].z - 10 * L[j].z + L[j + 1].z + L[j + 2].z + L[j + 3].z +
                             L[j + 4].z + L[j + 5].z;
            const double diffX = fx, diffY = fy, diffZ = fz;
            IdAndValue d;
            d.id_ = j;
            d.value_ = diffX * diffX + diffY * diffY + diffZ * diffZ;
            cloud_curvature.push_back(d);
        }
        for (int j = 0; j < 6; j++) {
            const int sector_length = total_points / 6;
            const int sector_start = sector_length * j;
            int sector_end = sector_length * (j + 1) - 1;
            if (j == 5) sector_end = total_points - 1;
            std::vector<IdAndValue> sub(cloud_curvature.begin() + sector_start, cloud_curvature.begin() + sector_end);
            LoamExtractFromSector(L, sub, order, edge, surf);
        }
    }
}

}  // namespace locref
