// loc_lib_amd/csrc/device_prims.hpp — the device-wide primitives the ingest-side kernels lean on (stable radix sort, exclusive sum,
// run-length encode), bound to rocPRIM directly. Two-phase calls as rocPRIM defines them: temp == nullptr returns the scratch size
// in `bytes`; the same call with a buffer of that size does the work on `s`.
#pragma once
#include <hip/hip_runtime.h>

#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_run_length_encode.hpp>
#include <rocprim/device/device_scan.hpp>

namespace locgpu {
namespace prim {

// LSD radix sort on bits [bit0, bit1) of the keys: stable, i.e. equal keys keep their input order (what every caller relies on).
template <class Key, class Value>
inline hipError_t sort_pairs(void* temp, size_t& bytes, const Key* keys_in, Key* keys_out, const Value* vals_in, Value* vals_out, size_t n, unsigned bit0, unsigned bit1,
                             hipStream_t s) {
    return rocprim::radix_sort_pairs(temp, bytes, keys_in, keys_out, vals_in, vals_out, n, bit0, bit1, s);
}

template <class Key>
inline hipError_t sort_keys(void* temp, size_t& bytes, const Key* keys_in, Key* keys_out, size_t n, unsigned bit0, unsigned bit1, hipStream_t s) {
    return rocprim::radix_sort_keys(temp, bytes, keys_in, keys_out, n, bit0, bit1, s);
}

// out[i] = in[0] + … + in[i-1]; in == out is allowed.
template <class T>
inline hipError_t exclusive_sum(void* temp, size_t& bytes, const T* in, T* out, size_t n, hipStream_t s) {
    return rocprim::exclusive_scan(temp, bytes, in, out, T(0), n, rocprim::plus<T>(), s);
}

// Runs of equal neighbours: unique_out[r], counts_out[r] for r < *n_runs_out.
template <class T, class Count, class Runs>
inline hipError_t run_length_encode(void* temp, size_t& bytes, const T* in, T* unique_out, Count* counts_out, Runs* n_runs_out, size_t n, hipStream_t s) {
    return rocprim::run_length_encode(temp, bytes, in, (unsigned int)n, unique_out, counts_out, n_runs_out, s);
}

}  // namespace prim
}  // namespace locgpu
