// oracle/locref_filters.hpp
//
// TEST INFRASTRUCTURE ONLY — CPU restatement of the cloud filters that sit either side of the matcher in the
// reference's front-ends. Nothing under loc_lib_amd/ or include/ uses this file.
//
// The reference's own code here is glue around PCL calls:
//   VoxelFilter::Filter   LocUtils/src/model/cloud_filter/voxel_filter.cpp:19-25   → pcl::VoxelGrid<PointXYZI>::filter
//   BoxFilter::Filter     LocUtils/src/model/cloud_filter/box_filter.cpp:25-32     → pcl::CropBox<PointXYZI>::filter
//                         (+ SetOrigin/CalculateEdge :53-66)
//   RemoveNanPoint        LocUtils/include/LocUtils/common/point_cloud_utils.h:13-20 → pcl::removeNaNFromPointCloud
//   Lio::AddCloud         LocUtils/src/slam/3d/lio.cpp:237-306 (local-map bookkeeping)
// PCL is a third-party dependency that is NOT under /root/reference. Pinned version: 1.8 (the prebuilt
// LocUtils/libs/libLocUtils.so NEEDs libpcl_filters.so.1.8; Ubuntu 18.04 ships 1.8.1). The algorithms restated below
// are the published ones of that release:
//   filters/include/pcl/filters/impl/voxel_grid.hpp  VoxelGrid<PointT>::applyFilter  (+ common/impl/centroid.hpp
//       CentroidPoint / AccumulatorXYZ / AccumulatorIntensity, common/impl/common.hpp getMinMax3D)
//   filters/include/pcl/filters/impl/crop_box.hpp    CropBox<PointT>::applyFilter(std::vector<int>&)
//   filters/include/pcl/filters/impl/filter.hpp      removeNaNFromPointCloud
// PARITY UNPINNED: the reference holds no test or golden vector for these calls, and PCL cannot be built here.
// One more thing is unpinned by construction: VoxelGrid sorts (voxel index, point index) pairs with std::sort and a
// comparison on the voxel index only, so the order in which a voxel's points are summed in float32 is whatever that
// unstable sort leaves. `order = 0` below does the same with this toolchain's std::sort (libstdc++'s introsort has not
// changed between GCC 7 and 11); `order = 1` sums in input order (what a stable sort gives) and is the order the GPU
// path uses. The two differ by float32 rounding of the sums only; tests bound that difference.
#pragma once
#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>

namespace locref {

struct PointXYZI { float x, y, z, intensity; };  // the fields of pcl::PointXYZI that the filters read or write

static inline bool is_finite_xyz(const PointXYZI& p) {  // pcl::isFinite / pcl_isfinite on x, y, z
    return std::isfinite(p.x) && std::isfinite(p.y) && std::isfinite(p.z);
}

// pcl::removeNaNFromPointCloud (filter.hpp): a dense cloud is copied unchanged — the flag is trusted, nothing is tested.
static inline size_t RemoveNaN(const PointXYZI* in, size_t n, bool is_dense, PointXYZI* out) {
    if (is_dense) {
        std::memcpy(out, in, n * sizeof(PointXYZI));
        return n;
    }
    size_t j = 0;
    for (size_t i = 0; i < n; ++i) {
        if (!is_finite_xyz(in[i])) continue;
        out[j++] = in[i];
    }
    return j;
}

// pcl::CropBox::applyFilter with identity transform, zero translation/rotation, negative_ = false (BoxFilter never
// sets them, box_filter.cpp:25-32). Bounds are inclusive; on a cloud flagged dense a NaN coordinate fails every `<`/`>`
// test and the point is KEPT.
static inline size_t CropBox(const PointXYZI* in, size_t n, bool is_dense, const float mn[3], const float mx[3], PointXYZI* out) {
    size_t j = 0;
    for (size_t i = 0; i < n; ++i) {
        const PointXYZI& p = in[i];
        if (!is_dense && !is_finite_xyz(p)) continue;
        if ((p.x < mn[0] || p.y < mn[1] || p.z < mn[2]) || (p.x > mx[0] || p.y > mx[1] || p.z > mx[2])) continue;
        out[j++] = p;
    }
    return j;
}

// BoxFilter::CalculateEdge (box_filter.cpp:59-66) with size_ = {-s, s, …} (:14-22): float32 sums.
static inline void BoxEdges(const float step[3], const float origin[3], float mn[3], float mx[3]) {
    for (int a = 0; a < 3; ++a) {
        mn[a] = -step[a] + origin[a];
        mx[a] = step[a] + origin[a];
    }
}

// pcl::transformPointCloud(in, out, Eigen::Matrix4d) as Lio::AddCloud calls it (lio.cpp:244,279): PCL 1.8's templated
// overload multiplies in double, left to right, and stores float; on a cloud not flagged dense it skips non-finite points
// (they stay as copied). m = row-major 3×4 [R | t].
static inline void TransformCloudF64(const PointXYZI* in, size_t n, bool is_dense, const double m[12], PointXYZI* out) {
    for (size_t i = 0; i < n; ++i) {
        out[i] = in[i];
        if (!is_dense && !is_finite_xyz(in[i])) continue;
        const double x = in[i].x, y = in[i].y, z = in[i].z;
        out[i].x = (float)(m[0] * x + m[1] * y + m[2] * z + m[3]);
        out[i].y = (float)(m[4] * x + m[5] * y + m[6] * z + m[7]);
        out[i].z = (float)(m[8] * x + m[9] * y + m[10] * z + m[11]);
    }
}

struct VoxelGridInfo {
    int status;        // 0 filtered; 1 leaf too small for the data (output = input, PCL warns); 2 empty input / no finite point
    int min_b[3], div_b[3];
    size_t n_voxels;
};

// pcl::VoxelGrid<PointXYZI>::applyFilter with the defaults VoxelFilter leaves in place (voxel_filter.cpp:11-25):
// leaf (v, v, v), downsample_all_data_ = true, min_points_per_voxel_ = 0, no filter field, save_leaf_layout_ = false.
// `out` needs room for n points. Returns the number of output points.
static inline size_t VoxelGrid(const PointXYZI* in, size_t n, bool is_dense, float leaf, int order, PointXYZI* out, VoxelGridInfo* info) {
    VoxelGridInfo local;
    VoxelGridInfo& I = info ? *info : local;
    I = VoxelGridInfo{};
    if (n == 0) { I.status = 2; return 0; }  // (PCL's arithmetic on an empty cloud is undefined; callers never pass one)
    const float inv = 1.0f / leaf;  // inverse_leaf_size_ = Array4f::Ones() / leaf_size_.array()

    // getMinMax3D(*input_, *indices_, min_p, max_p)
    float min_p[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, max_p[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
    for (size_t i = 0; i < n; ++i) {
        if (!is_dense && !is_finite_xyz(in[i])) continue;
        const float c[3] = {in[i].x, in[i].y, in[i].z};
        for (int a = 0; a < 3; ++a) {
            min_p[a] = c[a] < min_p[a] ? c[a] : min_p[a];
            max_p[a] = c[a] > max_p[a] ? c[a] : max_p[a];
        }
    }
    if (min_p[0] > max_p[0]) { I.status = 2; return 0; }  // not one finite point (PCL's arithmetic on ±FLT_MAX is undefined here)
    // "Leaf size is too small for the input dataset. Integer indices would overflow." → output = *input_
    const int64_t dx = (int64_t)((max_p[0] - min_p[0]) * inv) + 1, dy = (int64_t)((max_p[1] - min_p[1]) * inv) + 1,
                  dz = (int64_t)((max_p[2] - min_p[2]) * inv) + 1;
    if (dx * dy * dz > (int64_t)INT32_MAX) {
        std::memcpy(out, in, n * sizeof(PointXYZI));
        I.status = 1;
        return n;
    }
    int min_b[3], max_b[3], div_b[3];
    for (int a = 0; a < 3; ++a) {
        min_b[a] = (int)std::floor(min_p[a] * inv);
        max_b[a] = (int)std::floor(max_p[a] * inv);
        div_b[a] = max_b[a] - min_b[a] + 1;
        I.min_b[a] = min_b[a];
        I.div_b[a] = div_b[a];
    }
    const int mul[3] = {1, div_b[0], div_b[0] * div_b[1]};

    struct Entry {
        unsigned int idx, cloud_point_index;
        bool operator<(const Entry& o) const { return idx < o.idx; }  // cloud_point_index_idx::operator<
    };
    std::vector<Entry> index_vector;
    index_vector.reserve(n);
    for (size_t i = 0; i < n; ++i) {
        if (!is_dense && !is_finite_xyz(in[i])) continue;
        const int ijk0 = (int)(std::floor(in[i].x * inv) - (float)min_b[0]);
        const int ijk1 = (int)(std::floor(in[i].y * inv) - (float)min_b[1]);
        const int ijk2 = (int)(std::floor(in[i].z * inv) - (float)min_b[2]);
        const int idx = ijk0 * mul[0] + ijk1 * mul[1] + ijk2 * mul[2];
        index_vector.push_back(Entry{(unsigned int)idx, (unsigned int)i});
    }
    if (order == 0) std::sort(index_vector.begin(), index_vector.end());
    else std::stable_sort(index_vector.begin(), index_vector.end());

    size_t total = 0, index = 0;
    while (index < index_vector.size()) {
        size_t i = index + 1;
        while (i < index_vector.size() && index_vector[i].idx == index_vector[index].idx) ++i;
        // min_points_per_voxel_ = 0: every voxel is kept. CentroidPoint: float32 running sums, then one division each.
        float sx = 0.f, sy = 0.f, sz = 0.f, si = 0.f;
        for (size_t li = index; li < i; ++li) {
            const PointXYZI& p = in[index_vector[li].cloud_point_index];
            sx += p.x; sy += p.y; sz += p.z; si += p.intensity;
        }
        const float cnt = (float)(i - index);
        out[total++] = PointXYZI{sx / cnt, sy / cnt, sz / cnt, si / cnt};
        index = i;
    }
    I.n_voxels = total;
    return total;
}

// The keyframe branch of Lio::AddCloud (lio.cpp:268-306) and its first-frame branch (:238-256), without the matcher:
// what the local map (the next matching target) is after each keyframe. Clouds are already in the world frame.
struct LocalMap {
    size_t num_kfs;  // lio_option_.num_kfs_in_local_map_
    float leaf;      // lio_option_.local_map_filter_
    int order;       // VoxelGrid summation order, see above
    std::vector<std::vector<PointXYZI>> scans;  // scans_in_local_map_
    std::vector<bool> scan_dense;
    std::vector<PointXYZI> map;                  // local_map_
    bool map_dense = true;                       // a fresh pcl::PointCloud is dense; operator+= ANDs the flags
    int last_status = 0;

    void AddKeyframe(const PointXYZI* kf, size_t n, bool is_dense) {
        scans.emplace_back(kf, kf + n);
        scan_dense.push_back(is_dense);
        if (scans.size() > num_kfs) {          // :285-294 drop the oldest, rebuild from the retained (unfiltered) keyframes
            scans.erase(scans.begin());
            scan_dense.erase(scan_dense.begin());
            map.clear();
            map_dense = true;
            for (size_t k = 0; k < scans.size(); ++k) {
                map.insert(map.end(), scans[k].begin(), scans[k].end());
                map_dense = map_dense && scan_dense[k];
            }
        } else {                               // :295-298 append to the already filtered map
            map.insert(map.end(), kf, kf + n);
            map_dense = map_dense && is_dense;
        }
        std::vector<PointXYZI> filtered(map.size());
        VoxelGridInfo info;
        const size_t m = VoxelGrid(map.data(), map.size(), map_dense, leaf, order, filtered.data(), &info);  // :300, in place
        filtered.resize(m);
        map.swap(filtered);
        last_status = info.status;
        if (info.status == 0) map_dense = true;  // applyFilter sets output.is_dense; its pass-through copies the input's flag
    }
};

}  // namespace locref
