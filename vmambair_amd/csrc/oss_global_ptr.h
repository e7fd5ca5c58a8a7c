// Pointers that a kernel READS FROM MEMORY (descriptor tables of the grouped launches) instead of receiving as kernel arguments.
// The compiler knows that a kernel argument points to global memory; a pointer loaded from a table is "generic", and every access
// through it becomes a FLAT instruction: it counts on vmcnt AND lgkmcnt and may complete out of order with LDS operations, so each
// LDS wait of the kernel turns into `s_waitcnt vmcnt(0) lgkmcnt(0)` -- a prefetch window of 32 loads collapses to none (the grouped
// weight-gradient kernel ran at 1 TB/s for three rounds because of it; `tools/kernel_regs.py` / grep flat_load in the ISA).
// table_ptr() below reads the table entry as a 64-bit integer and makes a global pointer of it, so that the address-space inference
// pass rewrites every use as global_load / _store.  (Neither a cast of the generic pointer through address_space(1) and back nor
// `__builtin_assume(!is_shared && !is_private)` survives to that pass: both were tried, both left the FLAT instructions in place.)
#pragma once

namespace oss {

template <typename U, typename F>
__device__ __forceinline__ U *table_ptr(const F *field /* address of a pointer-typed member of a table entry in global memory */) {
    static_assert(sizeof(F) == 8, "a pointer member");
    unsigned long long raw;
    __builtin_memcpy(&raw, field, 8);
    typedef U __attribute__((address_space(1))) *GlobalPtr;
    return (U *)(GlobalPtr)raw;
}

}  // namespace oss
