// Shared pieces of the F-form strip convs of the 1024² level (conv_f16s_stripx.hip: four waves, one per SIMD; conv_f16s_stripx8.hip:
// eight waves, two per SIMD with specialised roles): LDS layout, the inline-assembly LDS accessors (see the comments there for why
// every LDS access of the loop is inline assembly), the kernel argument block.
#pragma once
#include "conv_common.hpp"
#include "sform.hpp"
#include <cstdint>
#include <cstdlib>

using namespace oodgan;

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void gbl_void;

#ifndef SX_ABL
#define SX_ABL 0
#endif
// cache policy of the 16-byte LDS-DMA pieces (raw rows, halo records, dot rows): 0 = default, 2 = nt (streaming)
#ifndef SX_NT_LD
#define SX_NT_LD 0
#endif
#ifndef SX_NT_ST
#define SX_NT_ST 0
#endif
#ifndef SX_PD_FWD
#define SX_PD_FWD 2
#endif

#ifdef OODGAN_CLOCK_STAMP
// Diagnostic build only (make STAMP=1): shader cycles a wave spends in the phases of the tile loop, summed over its tiles —
// [workgroup][wave][counted wait, barrier, matrix phase with the woven work, stores, whole loop, 100 MHz ticks of the loop]
extern __device__ unsigned long long* g_stripx_stamp;      // defined in conv_f16s_stripx.hip
extern __device__ long g_stripx_stamp_n;
#define SX_STAMP(v) const unsigned long long v = __builtin_amdgcn_s_memtime()
#else
#define SX_STAMP(v)
#endif

namespace {

constexpr int SX_C = 34;                               // records per ring row and channel block
constexpr int SX_ROW = 2 * SX_C * 64;                  // 4352 bytes: [kc 2][34 records][64 B]
constexpr int SX_GROUP = 4 * SX_ROW;                   // 17408: four rows
constexpr int SX_SMALL_ONE = 3072;
constexpr int SX_DOT_ONE = 16384;
// LDS layout for prefetch distance PD (tiles between the request of a group and its conversion): NG = PD + 3 ring groups — t-1 (its
// last two rows), t, t+1 (being converted) and PD in flight — and NB = PD + 1 buffers for everything else that travels with a batch.
// The backward (48 KiB of dot rows) has room for PD = 2; the forward would have room for 4, which measured the same as 2 (the
// counted wait and the barrier cost 0.1 us per tile in the stamp build: the prefetch is deep enough) — it keeps PD = 2 and
// leaves 60 KiB of LDS to kernels of the other HIP streams.
template <bool BWD>
struct SXL {
    static constexpr int PD = BWD ? 2 : SX_PD_FWD;
    static constexpr int NG = PD + 3, NB = PD + 1;
    static constexpr int RING = NG * SX_GROUP;
    static constexpr int HALO = RING;                              // NB x [4 waves][256 B]: raw halo records (columns 0 / 33) of a group
    static constexpr int SMALL = HALO + NB * 1024;                 // backward: NB x 3072 ([pixel slot][g_rgb 0..2, noise]); forward: NB x 1 KiB noise rows
    static constexpr int DOT = SMALL + NB * (BWD ? SX_SMALL_ONE : 1024);   // backward: NB x 16 KiB saved forward input of a tile, thread-private slots
    static constexpr int FIN = DOT + (BWD ? NB * SX_DOT_ONE : 0);  // final sums: [4 waves][32 ch][r,t] + [4][2] maxima
    static constexpr int CST = FIN + 4 * 32 * 2 * 4 + 64;          // [2 halves][16] epilogue scales, then [kc 2][quarter 4][7] float4 constants
    static constexpr int EPC = CST + 128 + 8 * 7 * 16;             // forward: [half 2][rr 4][bias, wr0, wr1, wr2][4 channels] floats
    static constexpr int SMEM = EPC + 512;                         // backward 151 KiB
};

#define SX_VM(n) ((((n) >> 4) & 3) << 14 | 0x0F70 | ((n) & 15))
#define SX_VML(n) ((((n) >> 4) & 3) << 14 | 0x0070 | ((n) & 15))      // ... and lgkmcnt(0)

// LDS reads of the loop as inline assembly.  A compiler-visible LDS load issued while an LDS-DMA is pending makes the waitcnt
// pass insert s_waitcnt vmcnt(0) in front of it (it cannot tell which DMA the load may alias), which would drain the two
// tiles of prefetch at every conversion and every epilogue.  The data read here has been waited for explicitly (counted vmcnt
// + barrier at the top of the tile).  Each block ends with its own lgkmcnt(0): the outputs are valid when it returns.
typedef float f32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ unsigned lds_addr(const void* p) {
    return (unsigned)reinterpret_cast<uintptr_t>((const __attribute__((address_space(3))) void*)p);
}
__device__ __forceinline__ void lds_read4(unsigned a, f32x4& r0, f32x4& r1, f32x4& r2, f32x4& r3) {
    asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %4 offset:1024\n\tds_read_b128 %2, %4 offset:2048\n\tds_read_b128 %3, %4 offset:3072\n\ts_waitcnt lgkmcnt(0)"
                 : "=&v"(r0), "=&v"(r1), "=&v"(r2), "=&v"(r3) : "v"(a) : "memory");
}
__device__ __forceinline__ void lds_read4_16(unsigned a, f32x4& r0, f32x4& r1, f32x4& r2, f32x4& r3) {
    asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %4 offset:16\n\tds_read_b128 %2, %4 offset:32\n\tds_read_b128 %3, %4 offset:48\n\ts_waitcnt lgkmcnt(0)"
                 : "=&v"(r0), "=&v"(r1), "=&v"(r2), "=&v"(r3) : "v"(a) : "memory");
}
__device__ __forceinline__ void lds_read7_16(unsigned a, f32x4 (&r)[7]) {
    asm volatile("ds_read_b128 %0, %7\n\tds_read_b128 %1, %7 offset:16\n\tds_read_b128 %2, %7 offset:32\n\tds_read_b128 %3, %7 offset:48\n\t"
                 "ds_read_b128 %4, %7 offset:64\n\tds_read_b128 %5, %7 offset:80\n\tds_read_b128 %6, %7 offset:96\n\ts_waitcnt lgkmcnt(0)"
                 : "=&v"(r[0]), "=&v"(r[1]), "=&v"(r[2]), "=&v"(r[3]), "=&v"(r[4]), "=&v"(r[5]), "=&v"(r[6]) : "v"(a) : "memory");
}
// the four rows of a thread's in-place conversion (row pitch 4352) and its halo record
__device__ __forceinline__ void lds_read_rows(unsigned a, unsigned ah, f32x4& r0, f32x4& r1, f32x4& r2, f32x4& r3, f32x4& r4) {
    asm volatile("ds_read_b128 %0, %5\n\tds_read_b128 %1, %5 offset:4352\n\tds_read_b128 %2, %5 offset:8704\n\tds_read_b128 %3, %5 offset:13056\n\t"
                 "ds_read_b128 %4, %6\n\ts_waitcnt lgkmcnt(0)"
                 : "=&v"(r0), "=&v"(r1), "=&v"(r2), "=&v"(r3), "=&v"(r4) : "v"(a), "v"(ah) : "memory");
}
// the per-pixel inputs of the same five units (pixel-slot pitch 16 B, 32 slots per row)
__device__ __forceinline__ void lds_read_small(unsigned a, unsigned ah, f32x4& r0, f32x4& r1, f32x4& r2, f32x4& r3, f32x4& r4) {
    asm volatile("ds_read_b128 %0, %5\n\tds_read_b128 %1, %5 offset:512\n\tds_read_b128 %2, %5 offset:1024\n\tds_read_b128 %3, %5 offset:1536\n\t"
                 "ds_read_b128 %4, %6\n\ts_waitcnt lgkmcnt(0)"
                 : "=&v"(r0), "=&v"(r1), "=&v"(r2), "=&v"(r3), "=&v"(r4) : "v"(a), "v"(ah) : "memory");
}
// LDS stores of the loop: the same reason (an LDS store while a DMA is pending gets vmcnt(0) in front).  They are ordered with
// the other LDS operations of the wave; the wait at the top of the next tile includes lgkmcnt(0) before its barrier.
__device__ __forceinline__ void lds_write16(unsigned a, unsigned x, unsigned y, unsigned z, unsigned w) {
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    const u32x4 v = {x, y, z, w};
    asm volatile("ds_write_b128 %0, %1" : : "v"(a), "v"(v) : "memory");
}
__device__ __forceinline__ void lds_write4(unsigned a, float v) { asm volatile("ds_write_b32 %0, %1" : : "v"(a), "v"(v) : "memory"); }
// Split form: the reads are ISSUED here and WAITED FOR by a later lds_wait*() that names their destinations as in/out operands —
// every consumer depends on that statement, so the matrix / VALU instructions between the two overlap the LDS latency.  (The
// destinations must not be touched in between; tools/check_stripx_isa.py verifies that the compiler did not insert a copy.)
__device__ __forceinline__ void lds_issue(f32x4& d, unsigned a) { asm volatile("ds_read_b128 %0, %1" : "=v"(d) : "v"(a) : "memory"); }
__device__ __forceinline__ void lds_issue(half8& d, unsigned a) { asm volatile("ds_read_b128 %0, %1" : "=v"(d) : "v"(a) : "memory"); }
__device__ __forceinline__ void lds_issue(float& d, unsigned a) { asm volatile("ds_read_b32 %0, %1" : "=v"(d) : "v"(a) : "memory"); }
__device__ __forceinline__ void lds_wait(half8 (&f)[6]) {
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(f[0]), "+v"(f[1]), "+v"(f[2]), "+v"(f[3]), "+v"(f[4]), "+v"(f[5]) : : "memory");
}
__device__ __forceinline__ void lds_wait(f32x4& a) { asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a) : : "memory"); }
__device__ __forceinline__ void lds_wait(f32x4& a, f32x4& b) { asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a), "+v"(b) : : "memory"); }
__device__ __forceinline__ void lds_wait(f32x4& a, f32x4& b, f32x4& c, f32x4& d) {
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : : "memory");
}
__device__ __forceinline__ void lds_wait(f32x4 (&r)[7]) {
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]) : : "memory");
}
__device__ __forceinline__ void lds_wait(float& a) { asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a) : : "memory"); }
__device__ __forceinline__ void lds_read2(unsigned a0, unsigned a1, f32x4& r0, f32x4& r1) {
    asm volatile("ds_read_b128 %0, %2\n\tds_read_b128 %1, %3\n\ts_waitcnt lgkmcnt(0)" : "=&v"(r0), "=&v"(r1) : "v"(a0), "v"(a1) : "memory");
}
__device__ __forceinline__ f32x4 lds_read1x4(unsigned a) {
    f32x4 r;
    asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=&v"(r) : "v"(a) : "memory");
    return r;
}
__device__ __forceinline__ float lds_read1(unsigned a) {
    float r;
    asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=&v"(r) : "v"(a) : "memory");
    return r;
}
__device__ __forceinline__ float lds_read_sum4(unsigned a) {       // four floats 128 B apart
    float r0, r1, r2, r3;
    asm volatile("ds_read_b32 %0, %4\n\tds_read_b32 %1, %4 offset:128\n\tds_read_b32 %2, %4 offset:256\n\tds_read_b32 %3, %4 offset:384\n\ts_waitcnt lgkmcnt(0)"
                 : "=&v"(r0), "=&v"(r1), "=&v"(r2), "=&v"(r3) : "v"(a) : "memory");
    return r0 + r1 + r2 + r3;
}

}  // namespace

namespace oodgan {

struct StripX {
    oodgan_conv_args a;
    oodgan_actbwd_fuse f;    // x_fform == 2
    const float* w_unscale;
    const void* zeros;       // >= 64 bytes of zeros in device memory
    int tiles_x, tiles_y, seg_tiles, nseg, Mp;     // tiles of FOUR rows
    long out_plane;
    int nparts;
};

// the eight-wave forward instance (conv_f16s_stripx8.hip); `p` is complete (conv_f16s_stripx.hip: launch_s1_stripx)
int launch_s1_stripx8_fwd(const StripX& p, const void* wpk16, hipStream_t st);
bool stripx8_init();

}  // namespace oodgan
