// launch_util.h -- host-side helpers shared by the launchers, and the XCD-aware workgroup order.
#pragma once
#include <hip/hip_runtime.h>
#include <atomic>

namespace bfsr {

// Raise a kernel's dynamic-LDS limit once per DEVICE (hipFuncSetAttribute is per device).  `done` is a per-kernel bit mask of
// devices already configured: an idempotent cache, safe under concurrent callers (a racing second call repeats the same
// attribute write), so the library stays re-entrant across host threads, streams and devices.
inline int ensure_dynamic_lds(const void* kernel, int bytes, std::atomic<unsigned long long>& done)
{
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return -1;
    const unsigned long long bit = 1ull << (dev & 63);
    if (done.load(std::memory_order_acquire) & bit) return 0;
    if (hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes) != hipSuccess) return -1;
    done.fetch_or(bit, std::memory_order_release);
    return 0;
}

// Compute-unit count of the CURRENT device, queried once per device (the persistent kernels size their grids with it on every
// launch: hundreds of RDB convs per step).  Returns -1 if the runtime cannot tell -- the caller fails the launch instead of
// guessing a grid.
inline int cu_count()
{
    static std::atomic<int> cache[64];              // zero-initialised; 0 = not queried yet
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return -1;
    if (dev < 0 || dev >= 64) {                     // beyond the cache: ask every time
        int n = 0;
        return hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n > 0 ? n : -1;
    }
    int n = cache[dev].load(std::memory_order_acquire);
    if (n > 0) return n;
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) return -1;
    cache[dev].store(n, std::memory_order_release);
    return n;
}

// XCD-aware workgroup order (MI355X: block b runs on XCD b % 8, each XCD has a private L2): returns the logical work index of
// hardware block `bid` such that every XCD walks one CONTIGUOUS range of logical indices -- neighbouring tiles (shared halo
// rows, the cout groups of one input tile) then hit the same L2.  Bijective for any grid size.
__device__ __forceinline__ unsigned xcd_order(unsigned bid, unsigned nblk)
{
    const unsigned xcd = bid & 7u, slot = bid >> 3;
    const unsigned q = nblk >> 3, r = nblk & 7u;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + slot;
}

// 16-byte buffer store with a wave-uniform byte offset on top of the per-lane one.  The uniform part is ADDED INTO THE VGPR OFFSET and
// the instruction's soffset stays the literal 0 on purpose.  Measured on gfx950 (round 4, tools/exp/h2x_twice.py):
// `buffer_store_dwordx4 v[14:17], v96, s[20:23], s5 offen` followed directly by `v_cvt_f32_f16 v16, v7` stored the NEW v16 in lanes
// 12-15 / 28-31 / 44-47 / 60-63 of one launch in ~10 -- the ">64-bit store data" hazard of the GFX9 manual, which the manual (and LLVM's
// hazard recognizer, GCNHazardRecognizer::createsVALUHazard) exempt when soffset is an SGPR.  With a non-register soffset hipcc pads the
// following VALU write itself.
typedef unsigned store_u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void store_b128(__amdgpu_buffer_rsrc_t rs, store_u32x4 data, unsigned lane_off, unsigned uniform_off)
{
    __builtin_amdgcn_raw_buffer_store_b128(data, rs, lane_off + uniform_off, 0, 0);
}
// The same store for the LARGE streamed outputs of the hoisted conditioning convs (quad-major fp32 tensors of 1024 channels: gigabytes per launch, read by
// a later launch at the earliest): cache policy BFSR_OUT_AUX (0 default; 2 = nt, a measurement build: tools/exp/build_nt.sh).
#ifndef BFSR_OUT_AUX
#define BFSR_OUT_AUX 0
#endif
__device__ __forceinline__ void store_b128_stream(__amdgpu_buffer_rsrc_t rs, store_u32x4 data, unsigned lane_off, unsigned uniform_off)
{
    __builtin_amdgcn_raw_buffer_store_b128(data, rs, lane_off + uniform_off, 0, BFSR_OUT_AUX);
}


// fp32 -> the fp32 value of its fp16 rounding, PINNED in a register: the hi term of the two-term fp16 split.  Without the pin hipcc may form the hi term twice in
// two different ways: where the value is a product, `(_Float16)(a * b)` for the subtraction that forms the lo term is contracted into v_fma_mixlo_f16 (ONE rounding
// of the exact product, in spite of -ffp-contract=off) while the stored hi plane is converted from the fp32-rounded product (v_cvt_pk_f16_f32: TWO roundings).  Where
// the two disagree (the fp32 product lands on an fp16 tie) hi + lo misses the value by an fp16 ulp -- found in round 5, when a compiler flag changed the instruction
// selection of the fused LINF MLP: 6e-7 -> 3.7e-5 against the CPU double (the two v_cvt instructions themselves agree on 2.7e8 inputs, tools/exp/cvt_probe.hip).
// (_Float16)pin_f16(v) is exact, so both uses see the same rounded value whatever the compiler picks.
__device__ __forceinline__ float pin_f16(float v)
{
    float hf = (float)(_Float16)v;
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("" : "+v"(hf));
#endif
    return hf;
}
}  // namespace bfsr
