// Error reporting / version / device probing for the C ABI.
#include "common.hpp"
#include <atomic>
#include <cstdlib>
#include <cstring>
#include <new>

namespace oodgan {
static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
}  // namespace oodgan

// ---- dispatch tunables: read from the environment ONCE (first use), changed afterwards only through oodgan_set_tunable.
// (Round 2 called getenv() on every conv launch of the hot path.)
namespace oodgan {
namespace {
struct Tunable { const char* name; const char* env; long def; std::atomic<long> val; std::atomic<int> init; };
Tunable g_tun[OODGAN_TUN_COUNT] = {
    {"s1_big_min_items", "OODGAN_S1_BIG_MIN_ITEMS", 128, {0}, {0}},
    {"s2_big_min_items", "OODGAN_S2_BIG_MIN_ITEMS", 128, {0}, {0}},
    {"t2_big_min_items", "OODGAN_T2_BIG_MIN_ITEMS", 128, {0}, {0}},
    {"blurt_strip", "OODGAN_BLURT_STRIP", 1, {0}, {0}},
    {"blur_strip", "OODGAN_BLUR_STRIP", 1, {0}, {0}},
    {"upvb_waves", "OODGAN_UPVB_WAVES", 12, {0}, {0}},
    {"fewout_quad", "OODGAN_FEWOUT_QUAD", 1, {0}, {0}},
    {"tiny_mid_max", "OODGAN_TINY_MID_MAX", 1024, {0}, {0}},
    {"stripx_waves", "OODGAN_STRIPX_WAVES", 4, {0}, {0}},
};
}  // namespace
long tunable(int id) {
    Tunable& t = g_tun[id];
    if (!t.init.load(std::memory_order_acquire)) {
        const char* e = getenv(t.env);
        t.val.store(e ? atol(e) : t.def, std::memory_order_relaxed);
        t.init.store(1, std::memory_order_release);
    }
    return t.val.load(std::memory_order_relaxed);
}
}  // namespace oodgan

extern "C" int oodgan_set_tunable(const char* name, long value) {
    OODGAN_REQUIRE(name != nullptr, "set_tunable: null name");
    for (int i = 0; i < oodgan::OODGAN_TUN_COUNT; ++i)
        if (strcmp(name, oodgan::g_tun[i].name) == 0) {
            oodgan::g_tun[i].val.store(value, std::memory_order_relaxed);
            oodgan::g_tun[i].init.store(1, std::memory_order_release);
            return OODGAN_OK;
        }
    oodgan::set_error("set_tunable: unknown tunable '%s'", name);
    return OODGAN_E_ARG;
}
extern "C" long oodgan_get_tunable(const char* name) {
    if (name)
        for (int i = 0; i < oodgan::OODGAN_TUN_COUNT; ++i)
            if (strcmp(name, oodgan::g_tun[i].name) == 0) return oodgan::tunable(i);
    return -1;
}

// ---- one device per process
namespace oodgan {
bool bound_device_ok(const char* what) {
    static std::atomic<int> bound{-1};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) {
        (void)hipGetLastError();
        set_error("%s: no current HIP device", what);
        return false;
    }
    int expected = -1;
    if (bound.compare_exchange_strong(expected, dev, std::memory_order_acq_rel) || expected == dev) return true;
    set_error("%s: liboodgan_hip.so is bound to HIP device %d (its first use in this process) but device %d is current — one process per GPU "
              "(select the device with HIP_VISIBLE_DEVICES or torch.cuda.set_device before the first call)", what, expected, dev);
    return false;
}
}  // namespace oodgan

// ---- dispatch counters: one relaxed atomic increment per conv call, on the host
namespace oodgan {
namespace {
const char* const g_dc_name[OODGAN_DC_COUNT] = {"stripx", "strip", "s1big", "s1v2", "s1pp", "tiny", "t2big", "t2v2", "t2gen", "s2big", "s2v2", "s2gen", "upvb", "s1big_ys", "s2big_fuse", "s2big_dotx_sform", "s1big_g2", "s2big_g2", "stripx_g2", "s2big_xh", "s1big_xh"};
std::atomic<long> g_dc[OODGAN_DC_COUNT];
}  // namespace
void count_dispatch(int id) { g_dc[id].fetch_add(1, std::memory_order_relaxed); }
}  // namespace oodgan
extern "C" long oodgan_dispatch_count(const char* name) {
    if (name)
        for (int i = 0; i < oodgan::OODGAN_DC_COUNT; ++i)
            if (strcmp(name, oodgan::g_dc_name[i]) == 0) return oodgan::g_dc[i].load(std::memory_order_relaxed);
    return -1;
}
extern "C" int oodgan_dispatch_reset(void) {
    for (int i = 0; i < oodgan::OODGAN_DC_COUNT; ++i) oodgan::g_dc[i].store(0, std::memory_order_relaxed);
    return OODGAN_OK;
}

namespace {
// zero fill as a plain kernel: a hipMemsetAsync captured into a hipGraph did not reproduce the eager result on replay (the memset node
// of this ROCm on private-pool memory; tests/test_encoder.py::test_graphed_forward_equals_eager) — a kernel node does
__global__ __launch_bounds__(256) void zero_kernel(unsigned char* __restrict__ p, long bytes) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    const long head = (16 - (reinterpret_cast<uintptr_t>(p) & 15)) & 15;        // bytes before the first 16-byte boundary
    const long h = head < bytes ? head : bytes;
    const long n16 = (bytes - h) >> 4;
    if (i < n16) reinterpret_cast<uint4*>(p + h)[i] = make_uint4(0u, 0u, 0u, 0u);
    if (i < h) p[i] = 0;
    const long tail = h + (n16 << 4);
    if (i < bytes - tail) p[tail + i] = 0;
}
}  // namespace

extern "C" int oodgan_zero(void* p, long bytes, void* stream) {
    OODGAN_REQUIRE(p != nullptr && bytes >= 0, "zero: bad args");
    if (bytes == 0) return OODGAN_OK;
    const long items = (bytes >> 4) + 16;
    OODGAN_REQUIRE((items + 255) / 256 < (1L << 31), "zero: too large");
    hipLaunchKernelGGL(zero_kernel, dim3((unsigned)((items + 255) / 256)), dim3(256), 0, oodgan::as_stream(stream),
                       reinterpret_cast<unsigned char*>(p), bytes);
    return oodgan::check_launch("zero");
}

// ---- launch plans (common.hpp: launch_rec)
#include <vector>
namespace oodgan {
struct PlanRec {
    std::vector<std::function<void()>> ops;
    bool recording = false;
};
thread_local PlanRec* tl_plan_rec = nullptr;
int g_null_launch = 0;
void plan_append(std::function<void()>&& f) { tl_plan_rec->ops.emplace_back(std::move(f)); }
}  // namespace oodgan

extern "C" void* oodgan_plan_create(void) { return new (std::nothrow) oodgan::PlanRec(); }
extern "C" int oodgan_plan_destroy(void* plan) {
    oodgan::PlanRec* p = static_cast<oodgan::PlanRec*>(plan);
    if (p && oodgan::tl_plan_rec == p) oodgan::tl_plan_rec = nullptr;
    delete p;
    return OODGAN_OK;
}
extern "C" int oodgan_plan_record_begin(void* plan) {
    oodgan::PlanRec* p = static_cast<oodgan::PlanRec*>(plan);
    OODGAN_REQUIRE(p != nullptr, "plan_record_begin: null plan");
    OODGAN_REQUIRE(oodgan::tl_plan_rec == nullptr, "plan_record_begin: this thread is already recording a plan");
    p->ops.clear();
    p->recording = true;
    oodgan::tl_plan_rec = p;
    return OODGAN_OK;
}
extern "C" long oodgan_plan_record_end(void* plan) {
    oodgan::PlanRec* p = static_cast<oodgan::PlanRec*>(plan);
    if (p == nullptr || oodgan::tl_plan_rec != p) {
        oodgan::set_error("plan_record_end: this thread is not recording that plan");
        return -1;
    }
    oodgan::tl_plan_rec = nullptr;
    p->recording = false;
    return (long)p->ops.size();
}
extern "C" long oodgan_plan_size(const void* plan) { return plan ? (long)static_cast<const oodgan::PlanRec*>(plan)->ops.size() : -1; }
extern "C" int oodgan_plan_run(void* plan, int times) {
    oodgan::PlanRec* p = static_cast<oodgan::PlanRec*>(plan);
    OODGAN_REQUIRE(p != nullptr && !p->recording && times >= 0, "plan_run: null plan, a plan still recording, or times < 0");
    OODGAN_REQUIRE(oodgan::tl_plan_rec == nullptr, "plan_run: this thread is recording a plan");
    for (int t = 0; t < times; ++t)
        for (const std::function<void()>& f : p->ops) f();
    return oodgan::check_launch("plan_run");
}
extern "C" int oodgan_plan_set_null_launch(int on) {
    oodgan::g_null_launch = on ? 1 : 0;
    return OODGAN_OK;
}

extern "C" int oodgan_version(void) { return 109; }      // 109: oodgan_conv_args gained x_hi_only (precision f16s-g2), oodgan_plan_*, LPIPS ops; 108: oodgan_align_input / oodgan_align_input_stats gained `diff` (AlignNet diff_fAndg=False), dispatch sub-counters, tunable stripx_waves; 107: oodgan_conv_args gained dotx_sform / dotx_scale; 106: oodgan_conv_args gained ys_vmax (+ rgb_y partial sums / ys from the 8-wave stride-1 kernel), round-4 helpers; 105: oodgan_upconv_vblur_fform, oodgan_zero; 104: oodgan_dispatch_count / oodgan_dispatch_reset; 102: oodgan_conv_args gained x_fform, dotx_fform, workspace, workspace_bytes; 103: oodgan_blur_act_sform_sep
extern "C" const char* oodgan_last_error(void) { return oodgan::g_err; }
extern "C" int oodgan_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) {
        (void)hipGetLastError();
        return 0;
    }
    return n;
}
