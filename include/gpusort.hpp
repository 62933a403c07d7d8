// gpusort.hpp -- header-only C++ shims over the C ABI (gpusort.h) that keep the call
// shapes of the reference's two sort entry points, so lsb/sort.cu- and
// msb/src/test.cu-shaped drivers compile against libgpusort.so with only the include
// and the cuda* -> hip* runtime names changed.
//
//   gpusort::DoubleBuffer<T>                      <- cub::DoubleBuffer<T>      lsb/cub/cub/util_type.cuh:785-817
//   gpusort::DeviceRadixSort::SortKeys/...        <- cub::DeviceRadixSort      lsb/cub/cub/device/device_radix_sort.cuh:248,595,754
//   gpusort::DeviceSegmentedRadixSort::Sort*      <- cub::DeviceSegmentedRadixSort lsb/cub/cub/device/device_segmented_radix_sort.cuh
//   gpusort::NullType                             <- cub::NullType
//   rdxsrt_unstable_sort<K,V,IndexT>(...)         <- msb/src/sort/gpu_radix_sort.h:197
//   rdxsrt_unstable_sort_keys / _pairs            <- msb/src/sort/gpu_radix_sort.h:511,544
//   RDXSRT_SortedSequence<K,V>                    <- msb/src/sort/gpu_radix_sort.h:31-34
//
// Supported key types: unsigned int, int, float (the graded configurations are u32) and,
// through the DoubleBuffer overloads, unsigned long long, long long, double; value type:
// any 4- or 8-byte trivially copyable type or NullType.  The plain-pointer (copy)
// overloads are 32-bit only.  rdxsrt_unstable_sort with 64-bit keys or values goes through the
// hybrid MSB path's wide kernel set (gs_msb_sort_wide), DeviceSegmentedRadixSort with them
// through gs_segmented_sort_wide.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstring>
#include <cstdio>
#include <type_traits>

#include "gpusort.h"

namespace gpusort {

struct NullType {};

template <typename T>
struct DoubleBuffer {
    T *d_buffers[2];
    int selector;
    DoubleBuffer() : d_buffers{nullptr, nullptr}, selector(0) {}
    DoubleBuffer(T *d_current, T *d_alternate) : d_buffers{d_current, d_alternate}, selector(0) {}
    T *Current() { return d_buffers[selector]; }
    T *Alternate() { return d_buffers[selector ^ 1]; }
};

template <typename K> struct KeyTraits;
template <> struct KeyTraits<unsigned int> { static constexpr int type = GS_KEY_U32; };
template <> struct KeyTraits<int> { static constexpr int type = GS_KEY_I32; };
template <> struct KeyTraits<float> { static constexpr int type = GS_KEY_F32; };
template <> struct KeyTraits<unsigned long long> { static constexpr int type = GS_KEY_U64; };
template <> struct KeyTraits<unsigned long> { static constexpr int type = sizeof(unsigned long) == 8 ? GS_KEY_U64 : GS_KEY_U32; };
template <> struct KeyTraits<long long> { static constexpr int type = GS_KEY_I64; };
template <> struct KeyTraits<long> { static constexpr int type = sizeof(long) == 8 ? GS_KEY_I64 : GS_KEY_I32; };
template <> struct KeyTraits<double> { static constexpr int type = GS_KEY_F64; };
// 8- and 16-bit keys (test_device_radix_sort.cu:1244-1250): through gs_lsb_sort_any
template <> struct KeyTraits<bool> { static constexpr int type = GS_KEY_U8; };
template <> struct KeyTraits<unsigned char> { static constexpr int type = GS_KEY_U8; };
template <> struct KeyTraits<signed char> { static constexpr int type = GS_KEY_I8; };
template <> struct KeyTraits<char> { static constexpr int type = (char)-1 < 0 ? GS_KEY_I8 : GS_KEY_U8; };
template <> struct KeyTraits<unsigned short> { static constexpr int type = GS_KEY_U16; };
template <> struct KeyTraits<short> { static constexpr int type = GS_KEY_I16; };

struct DeviceRadixSort {
    template <typename KeyT, typename ValueT>
    static hipError_t Dispatch(void *d_temp_storage, size_t &temp_storage_bytes, DoubleBuffer<KeyT> &d_keys,
                               DoubleBuffer<ValueT> *d_values, int num_items, int begin_bit, int end_bit,
                               bool descending, hipStream_t stream)
    {
        constexpr int VB = std::is_same<ValueT, NullType>::value ? 0 : (int)sizeof(ValueT);
        static_assert(sizeof(KeyT) == 1 || sizeof(KeyT) == 2 || sizeof(KeyT) == 4 || sizeof(KeyT) == 8, "8-, 16-, 32- or 64-bit keys");
        if constexpr (sizeof(KeyT) < 4 || (VB != 0 && VB != 4 && VB != 8)) {
            // 8- and 16-bit keys, values of any other size (1, 2, 16, ... bytes: TestBackend<KeyT, KeyT>, TestFoo): gs_lsb_sort_any.
            // It is the plain-pointer form underneath, so the result always lands in the ALTERNATE buffer and the
            // selector flips once.
            const int vb = d_values ? VB : 0;
            const size_t need_a = gs_lsb_any_temp_bytes((uint64_t)num_items, KeyTraits<KeyT>::type, vb);
            if (d_temp_storage == nullptr) {
                temp_storage_bytes = need_a;
                return hipSuccess;
            }
            const int sel_a = d_keys.selector;
            const int err_a = gs_lsb_sort_any(d_temp_storage, temp_storage_bytes, d_keys.d_buffers[sel_a], d_keys.d_buffers[sel_a ^ 1],
                                              d_values ? (const void *)d_values->d_buffers[sel_a] : nullptr,
                                              d_values ? (void *)d_values->d_buffers[sel_a ^ 1] : nullptr, (uint64_t)num_items,
                                              KeyTraits<KeyT>::type, vb, begin_bit, end_bit, descending ? 1 : 0, stream);
            if (err_a == 0 && num_items > 0) {
                d_keys.selector = sel_a ^ 1;
                if (d_values) d_values->selector = sel_a ^ 1;
            }
            return static_cast<hipError_t>(err_a);
        } else
        if constexpr (sizeof(KeyT) == 8 || VB == 8) {      // general kernels (gs_lsb_sort_wide)
            const size_t need_w = gs_lsb_wide_temp_bytes((uint64_t)num_items, (int)sizeof(KeyT), d_values ? VB : 0);
            if (d_temp_storage == nullptr) {
                temp_storage_bytes = need_w;
                return hipSuccess;
            }
            void *k2[2] = {d_keys.d_buffers[0], d_keys.d_buffers[1]};
            void *v2[2] = {nullptr, nullptr};
            if (d_values) { v2[0] = d_values->d_buffers[0]; v2[1] = d_values->d_buffers[1]; }
            int sel_w = d_keys.selector;
            const int err_w = gs_lsb_sort_wide(d_temp_storage, temp_storage_bytes, k2, d_values ? v2 : nullptr, &sel_w,
                                               (uint64_t)num_items, (int)sizeof(KeyT), d_values ? VB : 0, begin_bit,
                                               end_bit, descending ? 1 : 0, KeyTraits<KeyT>::type, stream);
            if (err_w == 0) {
                d_keys.selector = sel_w;
                if (d_values) d_values->selector = sel_w;
            }
            return static_cast<hipError_t>(err_w);
        }
        const size_t need = gs_lsb_temp_bytes((uint64_t)num_items, d_values != nullptr);
        if (d_temp_storage == nullptr) {            // size query (dispatch_radix_sort.cuh:1110)
            temp_storage_bytes = need;
            return hipSuccess;
        }
        uint32_t *keys[2] = {reinterpret_cast<uint32_t *>(d_keys.d_buffers[0]),
                             reinterpret_cast<uint32_t *>(d_keys.d_buffers[1])};
        uint32_t *vals[2] = {nullptr, nullptr};
        if (d_values) {
            vals[0] = reinterpret_cast<uint32_t *>(d_values->d_buffers[0]);
            vals[1] = reinterpret_cast<uint32_t *>(d_values->d_buffers[1]);
        }
        int sel = d_keys.selector;
        const int err = gs_lsb_sort_u32(d_temp_storage, temp_storage_bytes, keys, d_values ? vals : nullptr, &sel,
                                        (uint64_t)num_items, begin_bit, end_bit, descending ? 1 : 0,
                                        KeyTraits<KeyT>::type, stream);
        if (err == 0) {
            d_keys.selector = sel;
            if (d_values) d_values->selector = sel;
        }
        return static_cast<hipError_t>(err);
    }

    // plain-pointer overloads: input untouched, result in *_out (device_radix_sort.cuh:156-180, 503-527)
    template <typename KeyT, typename ValueT>
    static hipError_t DispatchCopy(void *d_temp_storage, size_t &temp_storage_bytes, const KeyT *d_keys_in,
                                   KeyT *d_keys_out, const ValueT *d_values_in, ValueT *d_values_out, int num_items,
                                   int begin_bit, int end_bit, bool descending, hipStream_t stream)
    {
        static_assert(sizeof(KeyT) == 4, "32-bit keys only");
        const bool pairs = d_values_in != nullptr;
        if (d_temp_storage == nullptr) {
            temp_storage_bytes = gs_lsb_copy_temp_bytes((uint64_t)num_items, pairs);
            return hipSuccess;
        }
        return static_cast<hipError_t>(gs_lsb_sort_copy_u32(
            d_temp_storage, temp_storage_bytes, reinterpret_cast<const uint32_t *>(d_keys_in),
            reinterpret_cast<uint32_t *>(d_keys_out), reinterpret_cast<const uint32_t *>(d_values_in),
            reinterpret_cast<uint32_t *>(d_values_out), (uint64_t)num_items, begin_bit, end_bit, descending ? 1 : 0,
            KeyTraits<KeyT>::type, stream));
    }
    template <typename KeyT>
    static hipError_t SortKeys(void *d_temp_storage, size_t &temp_storage_bytes, const KeyT *d_keys_in, KeyT *d_keys_out,
                               int num_items, int begin_bit = 0, int end_bit = sizeof(KeyT) * 8, hipStream_t stream = 0,
                               bool /*debug_synchronous*/ = false)
    {
        return DispatchCopy<KeyT, uint32_t>(d_temp_storage, temp_storage_bytes, d_keys_in, d_keys_out, nullptr, nullptr,
                                            num_items, begin_bit, end_bit, false, stream);
    }
    template <typename KeyT, typename ValueT>
    static hipError_t SortPairs(void *d_temp_storage, size_t &temp_storage_bytes, const KeyT *d_keys_in, KeyT *d_keys_out,
                                const ValueT *d_values_in, ValueT *d_values_out, int num_items, int begin_bit = 0,
                                int end_bit = sizeof(KeyT) * 8, hipStream_t stream = 0, bool /*debug_synchronous*/ = false)
    {
        static_assert(sizeof(ValueT) == 4, "32-bit values only");
        return DispatchCopy<KeyT, ValueT>(d_temp_storage, temp_storage_bytes, d_keys_in, d_keys_out, d_values_in,
                                          d_values_out, num_items, begin_bit, end_bit, false, stream);
    }

    template <typename KeyT>
    static hipError_t SortKeys(void *d_temp_storage, size_t &temp_storage_bytes, DoubleBuffer<KeyT> &d_keys,
                               int num_items, int begin_bit = 0, int end_bit = sizeof(KeyT) * 8,
                               hipStream_t stream = 0, bool /*debug_synchronous*/ = false)
    {
        return Dispatch<KeyT, NullType>(d_temp_storage, temp_storage_bytes, d_keys, nullptr, num_items, begin_bit,
                                        end_bit, false, stream);
    }
    template <typename KeyT>
    static hipError_t SortKeysDescending(void *d_temp_storage, size_t &temp_storage_bytes, DoubleBuffer<KeyT> &d_keys,
                                         int num_items, int begin_bit = 0, int end_bit = sizeof(KeyT) * 8,
                                         hipStream_t stream = 0, bool /*debug_synchronous*/ = false)
    {
        return Dispatch<KeyT, NullType>(d_temp_storage, temp_storage_bytes, d_keys, nullptr, num_items, begin_bit,
                                        end_bit, true, stream);
    }
    template <typename KeyT, typename ValueT>
    static hipError_t SortPairs(void *d_temp_storage, size_t &temp_storage_bytes, DoubleBuffer<KeyT> &d_keys,
                                DoubleBuffer<ValueT> &d_values, int num_items, int begin_bit = 0,
                                int end_bit = sizeof(KeyT) * 8, hipStream_t stream = 0,
                                bool /*debug_synchronous*/ = false)
    {
        return Dispatch<KeyT, ValueT>(d_temp_storage, temp_storage_bytes, d_keys, &d_values, num_items, begin_bit,
                                      end_bit, false, stream);
    }
    template <typename KeyT, typename ValueT>
    static hipError_t SortPairsDescending(void *d_temp_storage, size_t &temp_storage_bytes, DoubleBuffer<KeyT> &d_keys,
                                          DoubleBuffer<ValueT> &d_values, int num_items, int begin_bit = 0,
                                          int end_bit = sizeof(KeyT) * 8, hipStream_t stream = 0,
                                          bool /*debug_synchronous*/ = false)
    {
        return Dispatch<KeyT, ValueT>(d_temp_storage, temp_storage_bytes, d_keys, &d_values, num_items, begin_bit,
                                      end_bit, true, stream);
    }
};

// cub::DeviceSegmentedRadixSort, DoubleBuffer overloads (lsb/cub/cub/device/device_segmented_radix_sort.cuh:
// 266-289, 450-473, 607-629, 779-801): 32- or 64-bit keys, 32- or 64-bit values, int offsets.
struct DeviceSegmentedRadixSort {
    template <typename KeyT, typename ValueT>
    static hipError_t Dispatch(void *d_temp_storage, size_t &temp_storage_bytes, DoubleBuffer<KeyT> &d_keys,
                               DoubleBuffer<ValueT> *d_values, int num_items, int num_segments, const int *d_begin_offsets,
                               const int *d_end_offsets, int begin_bit, int end_bit, bool descending, hipStream_t stream)
    {
        constexpr bool keys_only = std::is_same<ValueT, NullType>::value;
        constexpr int KB = (int)sizeof(KeyT), VB = keys_only ? 0 : (int)sizeof(ValueT);
        static_assert(KB == 4 || KB == 8, "32- or 64-bit keys");
        static_assert(VB == 0 || VB == 4 || VB == 8, "32- or 64-bit values");
        constexpr bool wide = KB == 8 || VB == 8;
        const int vb = d_values ? VB : 0;
        const size_t need = wide ? gs_segmented_wide_temp_bytes((uint64_t)num_items, KB, vb, (uint32_t)num_segments)
                                 : gs_segmented_temp_bytes((uint64_t)num_items, d_values != nullptr, (uint32_t)num_segments);
        if (d_temp_storage == nullptr) {
            temp_storage_bytes = need;
            return hipSuccess;
        }
        int sel = d_keys.selector;
        int err;
        if constexpr (wide) {
            void *keys[2] = {d_keys.d_buffers[0], d_keys.d_buffers[1]};
            void *vals[2] = {nullptr, nullptr};
            if (d_values) { vals[0] = d_values->d_buffers[0]; vals[1] = d_values->d_buffers[1]; }
            err = gs_segmented_sort_wide(d_temp_storage, temp_storage_bytes, keys, d_values ? vals : nullptr, &sel, (uint64_t)num_items,
                                         (uint32_t)num_segments, d_begin_offsets, d_end_offsets, KB, vb, begin_bit, end_bit,
                                         descending ? 1 : 0, KeyTraits<KeyT>::type, stream);
        } else {
            uint32_t *keys[2] = {reinterpret_cast<uint32_t *>(d_keys.d_buffers[0]), reinterpret_cast<uint32_t *>(d_keys.d_buffers[1])};
            uint32_t *vals[2] = {nullptr, nullptr};
            if (d_values) {
                vals[0] = reinterpret_cast<uint32_t *>(d_values->d_buffers[0]);
                vals[1] = reinterpret_cast<uint32_t *>(d_values->d_buffers[1]);
            }
            err = gs_segmented_sort_u32(d_temp_storage, temp_storage_bytes, keys, d_values ? vals : nullptr, &sel, (uint64_t)num_items,
                                        (uint32_t)num_segments, d_begin_offsets, d_end_offsets, begin_bit, end_bit, descending ? 1 : 0,
                                        KeyTraits<KeyT>::type, stream);
        }
        if (err == 0) {
            d_keys.selector = sel;
            if (d_values) d_values->selector = sel;
        }
        return static_cast<hipError_t>(err);
    }
    template <typename KeyT>
    static hipError_t SortKeys(void *d_temp_storage, size_t &temp_storage_bytes, DoubleBuffer<KeyT> &d_keys, int num_items,
                               int num_segments, const int *d_begin_offsets, const int *d_end_offsets, int begin_bit = 0,
                               int end_bit = sizeof(KeyT) * 8, hipStream_t stream = 0, bool /*debug_synchronous*/ = false)
    {
        return Dispatch<KeyT, NullType>(d_temp_storage, temp_storage_bytes, d_keys, nullptr, num_items, num_segments,
                                        d_begin_offsets, d_end_offsets, begin_bit, end_bit, false, stream);
    }
    template <typename KeyT>
    static hipError_t SortKeysDescending(void *d_temp_storage, size_t &temp_storage_bytes, DoubleBuffer<KeyT> &d_keys,
                                         int num_items, int num_segments, const int *d_begin_offsets,
                                         const int *d_end_offsets, int begin_bit = 0, int end_bit = sizeof(KeyT) * 8,
                                         hipStream_t stream = 0, bool /*debug_synchronous*/ = false)
    {
        return Dispatch<KeyT, NullType>(d_temp_storage, temp_storage_bytes, d_keys, nullptr, num_items, num_segments,
                                        d_begin_offsets, d_end_offsets, begin_bit, end_bit, true, stream);
    }
    template <typename KeyT, typename ValueT>
    static hipError_t SortPairs(void *d_temp_storage, size_t &temp_storage_bytes, DoubleBuffer<KeyT> &d_keys,
                                DoubleBuffer<ValueT> &d_values, int num_items, int num_segments, const int *d_begin_offsets,
                                const int *d_end_offsets, int begin_bit = 0, int end_bit = sizeof(KeyT) * 8,
                                hipStream_t stream = 0, bool /*debug_synchronous*/ = false)
    {
        return Dispatch<KeyT, ValueT>(d_temp_storage, temp_storage_bytes, d_keys, &d_values, num_items, num_segments,
                                      d_begin_offsets, d_end_offsets, begin_bit, end_bit, false, stream);
    }
    template <typename KeyT, typename ValueT>
    static hipError_t SortPairsDescending(void *d_temp_storage, size_t &temp_storage_bytes, DoubleBuffer<KeyT> &d_keys,
                                          DoubleBuffer<ValueT> &d_values, int num_items, int num_segments,
                                          const int *d_begin_offsets, const int *d_end_offsets, int begin_bit = 0,
                                          int end_bit = sizeof(KeyT) * 8, hipStream_t stream = 0,
                                          bool /*debug_synchronous*/ = false)
    {
        return Dispatch<KeyT, ValueT>(d_temp_storage, temp_storage_bytes, d_keys, &d_values, num_items, num_segments,
                                      d_begin_offsets, d_end_offsets, begin_bit, end_bit, true, stream);
    }
};

}  // namespace gpusort

// ---- MSB driver shims (global namespace, as in the reference) ---------------------

template <typename KeyT, typename ValueT = gpusort::NullType>
struct RDXSRT_SortedSequence {
    KeyT *sorted_keys;
    ValueT *sorted_values;
};

// Scratch owner standing in for RDXSRT_GPUDataManager (gpu_radix_sort.h:50-166): one
// device allocation sized by gs_msb_temp_bytes instead of eight.
struct RDXSRT_GPUDataManager {
    void *d_temp = nullptr;
    size_t bytes = 0;
    RDXSRT_GPUDataManager(uint64_t key_count, bool has_values)
    {
        bytes = gs_msb_temp_bytes(key_count, has_values ? 1 : 0);
        if (hipMalloc(&d_temp, bytes ? bytes : 1) != hipSuccess) { d_temp = nullptr; bytes = 0; }
    }
    bool ok() const { return d_temp != nullptr; }   // false: the allocation failed; a sort with it returns {nullptr, nullptr}
    ~RDXSRT_GPUDataManager() { if (d_temp) (void)hipFree(d_temp); }
    RDXSRT_GPUDataManager(const RDXSRT_GPUDataManager &) = delete;
    RDXSRT_GPUDataManager &operator=(const RDXSRT_GPUDataManager &) = delete;
};

// Ascending, unstable; synchronous like the reference (gpu_radix_sort.h:489-491); dev_values == NULL
// (ValueT = NullType) sorts keys only (msb/src/test.cu:53).  Result pointers are returned and, for
// 32-bit keys, are the caller's input arrays.  The reference's signature has no error channel; a failure here
// (scratch allocation, a pre-allocated manager sized for fewer keys, a failed launch) is reported on stderr and
// returned as {nullptr, nullptr} -- never as pointers to unsorted data.
namespace gpusort {
inline void report_failure(const char *what, int err)
{
    std::fprintf(stderr, "gpusort: %s failed: hipError %d (%s)\n", what, err, gs_error_string(err));
}
}  // namespace gpusort
template <typename KeyT, typename ValueT, typename IndexT = unsigned int>
RDXSRT_SortedSequence<KeyT, ValueT> rdxsrt_unstable_sort(KeyT *dev_keys, ValueT *dev_values, IndexT key_count,
                                                         KeyT *dev_sorted_keys_out, ValueT *dev_sorted_values_out,
                                                         void * /*local_sort_configurations*/ = nullptr,
                                                         RDXSRT_GPUDataManager *pre_allocated_dm = nullptr,
                                                         hipStream_t stream = nullptr)
{
    constexpr bool keys_only = std::is_same<ValueT, gpusort::NullType>::value;
    const bool pairs = !keys_only && dev_values != nullptr;
    constexpr int VB = keys_only ? 0 : (int)sizeof(ValueT);
    static_assert(sizeof(KeyT) == 4 || sizeof(KeyT) == 8, "32- or 64-bit keys");
    static_assert(VB == 0 || VB == 4 || VB == 8, "32- or 64-bit values");
    if constexpr (sizeof(KeyT) == 8 || VB == 8) {
        // 64-bit keys and/or values: the wide kernel set of the hybrid MSB sort (gs_msb_sort_wide); result in the input arrays
        const int vb = pairs ? VB : 0;
        const size_t tb = gs_msb_wide_temp_bytes((uint64_t)key_count, (int)sizeof(KeyT), vb);
        void *temp = nullptr;
        int err = (int)hipMalloc(&temp, tb ? tb : 1);
        if (err != 0) {
            gpusort::report_failure("rdxsrt_unstable_sort: scratch allocation", err);
            return RDXSRT_SortedSequence<KeyT, ValueT>{nullptr, nullptr};
        }
        void *sk = nullptr, *sv = nullptr;
        err = gs_msb_sort_wide(temp, tb, dev_keys, pairs ? (void *)dev_values : nullptr, (uint64_t)key_count, dev_sorted_keys_out,
                               pairs ? (void *)dev_sorted_values_out : nullptr, (int)sizeof(KeyT), vb, &sk, &sv,
                               gpusort::KeyTraits<KeyT>::type, stream, 1);
        (void)hipFree(temp);
        if (err != 0) {
            gpusort::report_failure("rdxsrt_unstable_sort (64-bit path)", err);
            return RDXSRT_SortedSequence<KeyT, ValueT>{nullptr, nullptr};
        }
        return RDXSRT_SortedSequence<KeyT, ValueT>{reinterpret_cast<KeyT *>(sk), pairs ? reinterpret_cast<ValueT *>(sv) : nullptr};
    } else {
    RDXSRT_GPUDataManager *dm = pre_allocated_dm ? pre_allocated_dm : new RDXSRT_GPUDataManager((uint64_t)key_count, pairs);
    uint32_t *sk = nullptr, *sv = nullptr;
    const int err = dm->ok() ? gs_msb_sort_u32(dm->d_temp, dm->bytes, reinterpret_cast<uint32_t *>(dev_keys),
                                               pairs ? reinterpret_cast<uint32_t *>(dev_values) : nullptr, (uint64_t)key_count,
                                               reinterpret_cast<uint32_t *>(dev_sorted_keys_out),
                                               pairs ? reinterpret_cast<uint32_t *>(dev_sorted_values_out) : nullptr, &sk, &sv,
                                               gpusort::KeyTraits<KeyT>::type, stream, 1)
                             : (int)hipErrorOutOfMemory;
    if (!pre_allocated_dm) delete dm;
    if (err != 0) {
        gpusort::report_failure(err == (int)hipErrorOutOfMemory ? "rdxsrt_unstable_sort: scratch allocation" : "rdxsrt_unstable_sort", err);
        return RDXSRT_SortedSequence<KeyT, ValueT>{nullptr, nullptr};
    }
    return RDXSRT_SortedSequence<KeyT, ValueT>{reinterpret_cast<KeyT *>(sk), reinterpret_cast<ValueT *>(sv)};
    }
}

// Host-pointer conveniences (gpu_radix_sort.h:511-587): allocate, copy in, sort, copy the
// result back from the INPUT device arrays, free.  The reference's versions return void and check nothing; these keep
// the signature, check every runtime call, and on ANY failure (allocation, copy, sort) leave the output arrays
// UNTOUCHED and say why through report_failure -- the caller never receives stale or half-copied data as if it were sorted.
template <typename KeyT>
void rdxsrt_unstable_sort_keys(KeyT *keys, const unsigned long long key_count, KeyT *sorted_keys_out)
{
    KeyT *dev_keys = nullptr, *dev_keys_out = nullptr;
    const size_t bytes = sizeof(KeyT) * (key_count ? key_count : 1);
    int err = (int)hipMalloc(&dev_keys_out, bytes);
    if (!err) err = (int)hipMalloc(&dev_keys, bytes);
    if (!err) err = (int)hipMemcpy(dev_keys, keys, sizeof(KeyT) * key_count, hipMemcpyHostToDevice);
    if (err) {
        gpusort::report_failure("rdxsrt_unstable_sort_keys: device allocation / copy in", err);
    } else {
        auto seq = rdxsrt_unstable_sort<KeyT, gpusort::NullType, unsigned int>(dev_keys, nullptr, (unsigned int)key_count,
                                                                               dev_keys_out, nullptr);
        if (seq.sorted_keys) {          // (a failed sort has reported itself and returned {nullptr, nullptr})
            err = (int)hipMemcpy(sorted_keys_out, seq.sorted_keys, sizeof(KeyT) * key_count, hipMemcpyDeviceToHost);
            if (err) gpusort::report_failure("rdxsrt_unstable_sort_keys: copy out", err);
        }
    }
    (void)hipFree(dev_keys);
    (void)hipFree(dev_keys_out);
}

template <typename KeyT, typename ValueT>
void rdxsrt_unstable_sort_pairs(KeyT *keys, ValueT *values, const unsigned long long key_count, KeyT *sorted_keys_out,
                                ValueT *sorted_values_out)
{
    KeyT *dk = nullptr, *dko = nullptr;
    ValueT *dv = nullptr, *dvo = nullptr;
    const size_t kb = sizeof(KeyT) * (key_count ? key_count : 1), vb = sizeof(ValueT) * (key_count ? key_count : 1);
    int err = (int)hipMalloc(&dko, kb);
    if (!err) err = (int)hipMalloc(&dk, kb);
    if (!err) err = (int)hipMalloc(&dvo, vb);
    if (!err) err = (int)hipMalloc(&dv, vb);
    if (!err) err = (int)hipMemcpy(dk, keys, sizeof(KeyT) * key_count, hipMemcpyHostToDevice);
    if (!err) err = (int)hipMemcpy(dv, values, sizeof(ValueT) * key_count, hipMemcpyHostToDevice);
    if (err) {
        gpusort::report_failure("rdxsrt_unstable_sort_pairs: device allocation / copy in", err);
    } else {
        auto seq = rdxsrt_unstable_sort<KeyT, ValueT, unsigned int>(dk, dv, (unsigned int)key_count, dko, dvo);
        if (seq.sorted_keys && seq.sorted_values) {
            // both copies into scratch first would double the host memory; instead: keys, then values, and a failure of the
            // second is reported (the keys array then holds sorted keys whose values did not arrive -- said so)
            err = (int)hipMemcpy(sorted_keys_out, seq.sorted_keys, sizeof(KeyT) * key_count, hipMemcpyDeviceToHost);
            if (err) {
                gpusort::report_failure("rdxsrt_unstable_sort_pairs: copy out (nothing was written)", err);
            } else {
                err = (int)hipMemcpy(sorted_values_out, seq.sorted_values, sizeof(ValueT) * key_count, hipMemcpyDeviceToHost);
                if (err) gpusort::report_failure("rdxsrt_unstable_sort_pairs: copy out of the VALUES (the keys were written, the values were not)", err);
            }
        }
    }
    (void)hipFree(dk); (void)hipFree(dko); (void)hipFree(dv); (void)hipFree(dvo);
}
