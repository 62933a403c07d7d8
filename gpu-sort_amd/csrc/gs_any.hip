// gs_any.hip -- the rest of cub::DeviceRadixSort's type contract: 8- and 16-bit keys (bool / char / signed char /
// unsigned char / short / unsigned short) and values of ANY size (the 1- and 2-byte values of TestBackend<KeyT, KeyT>,
// the 16-byte TestFoo), lsb/cub/test/test_device_radix_sort.cu:930-945,1244-1265, lsb/cub/test/test_util.h:1004-1010.
//
// Not a kernel set of its own: the keys are mapped to order-preserving unsigned sort keys (the same twiddles as the 32-bit
// path, cub::Traits<K>::TwiddleIn, util_type.cuh:966-1089) next to their indices, the (sort key, index) pairs go through the
// STABLE LSB sort of this library (gs_lsb_sort_u32; gs_lsb_sort_wide for 64-bit keys) on [begin_bit, end_bit), and one gather
// moves the original keys and the values -- whatever their size -- into the order the indices came out in.  Keys only with
// narrow keys: the sort keys alone are sorted and narrowed back.  Stable, ascending / descending, bit sub-ranges of the
// key's own width; input arrays untouched, result in the output arrays (the plain-pointer form of the reference's API).
// Cost: one pass to build the sort keys, a 32-bit pairs sort (80 B per element over four passes; two passes for 16-bit keys,
// one for 8-bit keys: only the passes that hold key bits run), one gather.
#include "gs_device.hpp"
#include "gs_host.hpp"

namespace gs {

static inline size_t any_align(size_t x) { return (x + 255) & ~(size_t)255; }
static inline int any_key_bytes(int key_type)
{
    switch (key_type) {
    case GS_KEY_U8: case GS_KEY_I8: return 1;
    case GS_KEY_U16: case GS_KEY_I16: return 2;
    case GS_KEY_U32: case GS_KEY_I32: case GS_KEY_F32: return 4;
    case GS_KEY_U64: case GS_KEY_I64: case GS_KEY_F64: return 8;
    default: return 0;
    }
}

// narrow key -> order-preserving u32 (zero-extended: the key's bit b stays bit b, so [begin_bit, end_bit) means the same)
template <int KB>
__device__ __forceinline__ uint32_t any_load_sortkey(const void *keys, uint64_t i, int key_type)
{
    uint32_t k;
    if (KB == 1) k = reinterpret_cast<const uint8_t *>(keys)[i];
    else if (KB == 2) k = reinterpret_cast<const uint16_t *>(keys)[i];
    else k = reinterpret_cast<const uint32_t *>(keys)[i];
    if (key_type == GS_KEY_I8) k ^= 0x80u;
    else if (key_type == GS_KEY_I16) k ^= 0x8000u;
    else if (key_type == GS_KEY_I32) k ^= 0x80000000u;
    else if (key_type == GS_KEY_F32) k = twiddle_in(k, 1, 0u);
    return k;
}

template <int KB>
__global__ __launch_bounds__(256) void any_prepare_kernel(const void *__restrict__ keys, uint32_t *__restrict__ sortkeys,
                                                          uint32_t *__restrict__ idx, uint64_t n, int key_type)
{
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        sortkeys[i] = any_load_sortkey<KB>(keys, i, key_type);
        if (idx) idx[i] = (uint32_t)i;
    }
}

__global__ __launch_bounds__(256) void any_prepare64_kernel(const uint64_t *__restrict__ keys, uint64_t *__restrict__ sortkeys,
                                                            uint32_t *__restrict__ idx, uint64_t n)
{
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        sortkeys[i] = keys[i];          // gs_lsb_sort_wide applies the 64-bit twiddles itself
        idx[i] = (uint32_t)i;
    }
}

// keys only, narrow keys: sorted sort keys -> keys of the caller's type
template <int KB>
__global__ __launch_bounds__(256) void any_narrow_kernel(const uint32_t *__restrict__ sortkeys, void *__restrict__ keys_out, uint64_t n,
                                                         int key_type)
{
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        uint32_t k = sortkeys[i];
        if (key_type == GS_KEY_I8) k ^= 0x80u;
        else if (key_type == GS_KEY_I16) k ^= 0x8000u;
        if (KB == 1) reinterpret_cast<uint8_t *>(keys_out)[i] = (uint8_t)k;
        else reinterpret_cast<uint16_t *>(keys_out)[i] = (uint16_t)k;
    }
}

template <int B> struct AnyChunk;
template <> struct AnyChunk<1> { typedef uint8_t type; };
template <> struct AnyChunk<2> { typedef uint16_t type; };
template <> struct AnyChunk<4> { typedef uint32_t type; };
template <> struct AnyChunk<8> { typedef uint64_t type; };
template <> struct AnyChunk<16> { typedef uint4 type; };

// out[i] = in[idx[i]] for elements of EB bytes (EB = 0: any size, byte by byte)
template <int EB>
__device__ __forceinline__ void any_move(const void *in, void *out, uint64_t src, uint64_t dst, uint32_t bytes)
{
    if constexpr (EB == 0) {
        const uint8_t *p = reinterpret_cast<const uint8_t *>(in) + src * bytes;
        uint8_t *q = reinterpret_cast<uint8_t *>(out) + dst * bytes;
        for (uint32_t b = 0; b < bytes; ++b) q[b] = p[b];
    } else {
        typedef typename AnyChunk<EB>::type T;
        reinterpret_cast<T *>(out)[dst] = reinterpret_cast<const T *>(in)[src];
    }
}

template <int KB, int VB>
__global__ __launch_bounds__(256) void any_gather_kernel(const uint32_t *__restrict__ idx, const void *__restrict__ keys_in,
                                                         void *__restrict__ keys_out, const void *__restrict__ vals_in,
                                                         void *__restrict__ vals_out, uint64_t n, uint32_t val_bytes)
{
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const uint64_t s = idx[i];
        any_move<KB>(keys_in, keys_out, s, i, (uint32_t)KB);
        any_move<VB>(vals_in, vals_out, s, i, val_bytes);
    }
}

static inline dim3 any_grid(uint64_t n)
{
    const uint64_t b = (n + 255) / 256;
    return dim3((unsigned)(b < 1 ? 1 : (b > 65536 ? 65536 : b)));
}

template <int KB>
static void launch_gather(int vb, const uint32_t *idx, const void *ki, void *ko, const void *vi, void *vo, uint64_t n, hipStream_t s)
{
    const dim3 g = any_grid(n), b(256);
#define GS_G(VB_) hipLaunchKernelGGL((any_gather_kernel<KB, VB_>), g, b, 0, s, idx, ki, ko, vi, vo, n, (uint32_t)vb)
    switch (vb) {
    case 1: GS_G(1); break;
    case 2: GS_G(2); break;
    case 4: GS_G(4); break;
    case 8: GS_G(8); break;
    case 16: GS_G(16); break;
    default: GS_G(0); break;
    }
#undef GS_G
}

}  // namespace gs

using namespace gs;

extern "C" {

size_t gs_lsb_any_temp_bytes(uint64_t num_items, int key_type, int val_bytes)
{
    const int kb = any_key_bytes(key_type);
    if (kb == 0) return 0;
    const size_t n = (size_t)num_items;
    if (kb == 8)   // (u64 sort key, u32 index) through the wide sort: 2 x keys, 2 x indices
        return any_align(gs_lsb_wide_temp_bytes(num_items, 8, 4)) + 2 * any_align(n * 8) + 2 * any_align(n * 4);
    const bool pairs = val_bytes != 0 || kb == 4;    // 32-bit keys come here only for odd value sizes
    return any_align(gs_lsb_temp_bytes(num_items, pairs)) + 2 * any_align(n * 4) + (pairs ? 2 * any_align(n * 4) : 0);
}

int gs_lsb_sort_any(void *d_temp, size_t temp_bytes, const void *d_keys_in, void *d_keys_out, const void *d_vals_in,
                    void *d_vals_out, uint64_t num_items, int key_type, int val_bytes, int begin_bit, int end_bit, int descending,
                    void *stream)
{
    GS_CLEAR_STALE_ERROR();
    const int kb = any_key_bytes(key_type);
    if (kb == 0 || val_bytes < 0 || val_bytes > 4096) return hipErrorInvalidValue;
    if (begin_bit < 0 || end_bit > 8 * kb || begin_bit > end_bit) return hipErrorInvalidValue;
    if (num_items >= (1ull << 32)) return hipErrorInvalidValue;
    if (num_items == 0) return hipSuccess;              // (empty arrays may come with null pointers)
    if ((val_bytes != 0) != (d_vals_in != nullptr) || (val_bytes != 0) != (d_vals_out != nullptr)) return hipErrorInvalidValue;
    if (!d_keys_in || !d_keys_out || d_keys_in == d_keys_out || (val_bytes && d_vals_in == d_vals_out)) return hipErrorInvalidValue;
    if (val_bytes == 16 && (((uintptr_t)d_vals_in | (uintptr_t)d_vals_out) & 15u)) return hipErrorInvalidValue;   // 16-byte moves
    if (!d_temp || temp_bytes < gs_lsb_any_temp_bytes(num_items, key_type, val_bytes)) return hipErrorInvalidValue;
    hipStream_t s = (hipStream_t)stream;
    const uint64_t n = num_items;
    const dim3 g = any_grid(n), b(256);
    char *c = (char *)d_temp;

    if (kb == 8) {
        const size_t sort_ws = any_align(gs_lsb_wide_temp_bytes(n, 8, 4));
        uint64_t *sk[2] = {(uint64_t *)(c + sort_ws), (uint64_t *)(c + sort_ws + any_align(n * 8))};
        uint32_t *ix[2] = {(uint32_t *)(c + sort_ws + 2 * any_align(n * 8)), (uint32_t *)(c + sort_ws + 2 * any_align(n * 8) + any_align(n * 4))};
        hipLaunchKernelGGL(any_prepare64_kernel, g, b, 0, s, (const uint64_t *)d_keys_in, sk[0], ix[0], n);
        void *k2[2] = {sk[0], sk[1]}, *v2[2] = {ix[0], ix[1]};
        int sel = 0;
        const int e = gs_lsb_sort_wide(d_temp, sort_ws, k2, v2, &sel, n, 8, 4, begin_bit, end_bit, descending, key_type, s);
        if (e) return e;
        launch_gather<8>(val_bytes, ix[sel], d_keys_in, d_keys_out, d_vals_in, d_vals_out, n, s);
        return (int)hipGetLastError();
    }

    const bool pairs = val_bytes != 0 || kb == 4;
    const size_t sort_ws = any_align(gs_lsb_temp_bytes(n, pairs));
    uint32_t *sk[2] = {(uint32_t *)(c + sort_ws), (uint32_t *)(c + sort_ws + any_align(n * 4))};
    uint32_t *ix[2] = {nullptr, nullptr};
    if (pairs) {
        ix[0] = (uint32_t *)(c + sort_ws + 2 * any_align(n * 4));
        ix[1] = (uint32_t *)(c + sort_ws + 3 * any_align(n * 4));
    }
    if (kb == 1) hipLaunchKernelGGL(any_prepare_kernel<1>, g, b, 0, s, d_keys_in, sk[0], ix[0], n, key_type);
    else if (kb == 2) hipLaunchKernelGGL(any_prepare_kernel<2>, g, b, 0, s, d_keys_in, sk[0], ix[0], n, key_type);
    else hipLaunchKernelGGL(any_prepare_kernel<4>, g, b, 0, s, d_keys_in, sk[0], ix[0], n, key_type);
    int sel = 0;
    // the sort keys are twiddled already: plain unsigned keys from here on; descending = the complement inside the sort
    const int e = gs_lsb_sort_u32(d_temp, sort_ws, sk, pairs ? ix : nullptr, &sel, n, begin_bit, end_bit, descending, GS_KEY_U32, s);
    if (e) return e;
    if (!pairs) {
        if (kb == 1) hipLaunchKernelGGL(any_narrow_kernel<1>, g, b, 0, s, (const uint32_t *)sk[sel], d_keys_out, n, key_type);
        else hipLaunchKernelGGL(any_narrow_kernel<2>, g, b, 0, s, (const uint32_t *)sk[sel], d_keys_out, n, key_type);
    } else if (kb == 1) {
        launch_gather<1>(val_bytes, ix[sel], d_keys_in, d_keys_out, d_vals_in, d_vals_out, n, s);
    } else if (kb == 2) {
        launch_gather<2>(val_bytes, ix[sel], d_keys_in, d_keys_out, d_vals_in, d_vals_out, n, s);
    } else {
        launch_gather<4>(val_bytes, ix[sel], d_keys_in, d_keys_out, d_vals_in, d_vals_out, n, s);
    }
    return (int)hipGetLastError();
}

}  // extern "C"
