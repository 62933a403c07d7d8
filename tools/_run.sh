tools/compare/rocprim_sort 30 2>&1 | tail -4
tools/compare/rocprim_sort 30 pairs 2>&1 | tail -5
