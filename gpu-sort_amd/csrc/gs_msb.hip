// gs_msb.hip -- placeholder until the MSB hybrid path lands (next commit).
#include "gs_device.hpp"
#include "gs_host.hpp"

extern "C" {
size_t gs_msb_temp_bytes(uint64_t, int) { return 0; }
int gs_msb_sort_u32(void *, size_t, uint32_t *, uint32_t *, uint64_t, uint32_t *, uint32_t *, uint32_t **, uint32_t **,
                    int, void *, int) { return hipErrorNotSupported; }
int gs_shard_histogram_u32(const uint32_t *, uint64_t, int, uint64_t *, int, void *) { return hipErrorNotSupported; }
int gs_shard_partition_u32(void *, size_t, const uint32_t *, uint32_t *, const uint32_t *, uint32_t *, uint64_t, int,
                           const uint8_t *, int, uint64_t *, int, void *) { return hipErrorNotSupported; }
}
