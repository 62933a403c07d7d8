"""MI355X-native radix sort: Python host mirror of the reference's entry points.

The compute path is hand-written HIP for gfx950 behind the C ABI of
include/gpusort.h (libgpusort.so); this package only carries device pointers
and streams to it.  torch is used for device memory, streams and
torch.distributed -- plumbing, not the product.

Names follow the reference:
  DoubleBuffer, DeviceRadixSort.SortKeys/SortPairs/SortKeysDescending/
  SortPairsDescending           (lsb/cub/cub/device/device_radix_sort.cuh)
  sortKeysGPU / sortPairsGPU    (lsb/sort.cu:25-76)
  rdxsrt_unstable_sort, rdxsrt_unstable_sort_keys/_pairs, RDXSRT_SortedSequence
                                (msb/src/sort/gpu_radix_sort.h:31-34,197,511,544)
"""
from ._lib import (GpuSortError, GS_KEY_U32, GS_KEY_I32, GS_KEY_F32, GS_KEY_U64, GS_KEY_I64, GS_KEY_F64, GS_KEY_U8, GS_KEY_I8, GS_KEY_U16, GS_KEY_I16, GS_GEN_UNIFORM, GS_GEN_ZIPF,
                   GS_GEN_ENTROPY_AND, GS_GEN_ENUMERATED, LIB_PATH, lib, KernelProfile)
from .lsb import DoubleBuffer, DeviceRadixSort, DeviceSegmentedRadixSort, sortKeysGPU, sortPairsGPU, lsb_pass_kernels
from .datagen import (generate_random_keys, generate_uniform_keys, generate_zipf_keys, generate_enumerated_values,
                      check_sorted, check_pairs_enumerated)
from .msb import RDXSRT_SortedSequence, rdxsrt_unstable_sort, rdxsrt_unstable_sort_keys, rdxsrt_unstable_sort_pairs

__all__ = [n for n in dir() if not n.startswith("_")]
