// hip_emu.h — a small CPU stand-in for the HIP device model.  TEST INFRASTRUCTURE ONLY.
//
// There is no GPU in the build container, and GPU minutes are scarce, so the `-m "not gpu"`
// suite compiles the *same* kernel and host sources of mimic3_amd/csrc with g++ against this
// header (-DMI355_EMU) into tests/emu/libmi355vits_emu.so and checks their logic (indexing,
// LDS tiling, barriers, wave shuffles, MFMA fragment layouts, host orchestration) against the
// oracle at small sizes.  It is never loaded by the product path: mimic3_amd/_native.py only
// opens mimic3_amd/csrc/libmi355vits.so (hipcc, gfx950) and raises if that is missing.
//
// Model: every workgroup runs on one OS thread; its work-items are cooperative fibers that
// run until they reach a workgroup barrier or a wave-level operation (shuffle/ballot/MFMA),
// which are rendezvous points.  Wave = 64 lanes (gfx950).  MFMA lane<->element maps follow
// /opt/skills/guides/cdna_hip_programming.md §3 and accumulate as a k-ordered fmaf chain.
// A missing barrier shows up as a wrong result (a fiber runs ahead until its next rendezvous);
// HIPEMU_REVERSE=1 runs fibers in reverse order to catch the opposite direction.
#pragma once
#include <atomic>
#include <chrono>
#include <cmath>
#include <condition_variable>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>
#include <sys/mman.h>

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
#define __launch_bounds__(...)
#define __shared__ static thread_local

struct dim3 {
    unsigned x, y, z;
    dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
struct float2 { float x, y; };
struct float4 { float x, y, z, w; };
struct int2 { int x, y; };
struct uint4 { unsigned x, y, z, w; };
struct uint2 { unsigned x, y; };
struct int4 { int x, y, z, w; };
static inline float4 make_float4(float x, float y, float z, float w) { return float4{x, y, z, w}; }
static inline float2 make_float2(float x, float y) { return float2{x, y}; }

typedef int hipError_t;
typedef void* hipStream_t;
struct hipemu_event { std::chrono::steady_clock::time_point t; };
typedef hipemu_event* hipEvent_t;
enum { hipSuccess = 0, hipErrorInvalidValue = 1, hipErrorOutOfMemory = 2 };
enum hipMemcpyKind { hipMemcpyHostToDevice, hipMemcpyDeviceToHost, hipMemcpyDeviceToDevice, hipMemcpyHostToHost, hipMemcpyDefault };
enum { hipStreamNonBlocking = 1, hipHostMallocDefault = 0, hipEventDefault = 0 };

static inline const char* hipGetErrorString(hipError_t e) { return e == hipSuccess ? "hipSuccess" : "hipemu error"; }
static inline hipError_t hipGetLastError() { return hipSuccess; }
static inline hipError_t hipSetDevice(int) { return hipSuccess; }
static inline hipError_t hipGetDevice(int* d) { *d = 0; return hipSuccess; }
// MI355_EMU_DEVICES=N makes the CPU model report N identical "devices" (multi-device host logic tests)
static inline hipError_t hipGetDeviceCount(int* n) {
    const char* e = getenv("MI355_EMU_DEVICES");
    const int v = e ? atoi(e) : 1;
    *n = v >= 1 && v <= 64 ? v : 1;
    return hipSuccess;
}
static inline hipError_t hipDeviceSynchronize() { return hipSuccess; }
static inline hipError_t hipMalloc(void** p, size_t n) {
    *p = nullptr;
    if (posix_memalign(p, 256, n ? n : 256)) return hipErrorOutOfMemory;
    memset(*p, 0xCD, n);  // poison: reading uninitialised device memory shows up as garbage
    return hipSuccess;
}
static inline hipError_t hipFree(void* p) { free(p); return hipSuccess; }
static inline hipError_t hipHostMalloc(void** p, size_t n, unsigned = 0) { return hipMalloc(p, n); }
static inline hipError_t hipHostFree(void* p) { free(p); return hipSuccess; }
static inline hipError_t hipMemcpy(void* d, const void* s, size_t n, hipMemcpyKind) { memcpy(d, s, n); return hipSuccess; }
static inline hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind, hipStream_t) { memcpy(d, s, n); return hipSuccess; }
static inline hipError_t hipMemset(void* d, int v, size_t n) { memset(d, v, n); return hipSuccess; }
static inline hipError_t hipMemsetAsync(void* d, int v, size_t n, hipStream_t) { memset(d, v, n); return hipSuccess; }
static inline hipError_t hipStreamCreate(hipStream_t* s) { *s = nullptr; return hipSuccess; }
static inline hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned) { *s = nullptr; return hipSuccess; }
static inline hipError_t hipStreamDestroy(hipStream_t) { return hipSuccess; }
static inline hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
static inline hipError_t hipEventCreate(hipEvent_t* e) { *e = new hipemu_event(); return hipSuccess; }
static inline hipError_t hipEventDestroy(hipEvent_t e) { delete e; return hipSuccess; }
static inline hipError_t hipEventRecord(hipEvent_t e, hipStream_t) { e->t = std::chrono::steady_clock::now(); return hipSuccess; }
static inline hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
static inline hipError_t hipEventElapsedTime(float* ms, hipEvent_t a, hipEvent_t b) {
    *ms = std::chrono::duration<float, std::milli>(b->t - a->t).count();
    return hipSuccess;
}
struct hipDeviceProp_t { char name[256]; int multiProcessorCount; size_t totalGlobalMem; char gcnArchName[256]; };
static inline hipError_t hipGetDeviceProperties(hipDeviceProp_t* p, int) {
    memset(p, 0, sizeof(*p));
    strcpy(p->name, "hipemu (CPU)");
    strcpy(p->gcnArchName, "emu");
    p->multiProcessorCount = 8;
    p->totalGlobalMem = size_t(8) << 30;
    return hipSuccess;
}

namespace hipemu {

constexpr int WAVE = 64;
constexpr int MAX_THREADS = 1024;
constexpr int MAX_WAVES = MAX_THREADS / WAVE;
constexpr size_t STACK_BYTES = 256 * 1024;
constexpr size_t DYN_SMEM_BYTES = 160 * 1024;

extern "C" void hipemu_switch(void** from_sp, void* to_sp);

enum State { READY = 0, WAIT_BLOCK = 1, WAIT_WAVE = 2, DONE = 3 };

struct Fiber {
    void* sp;
    dim3 tid;
    int lin;
    int state;
    unsigned wave_ops;
    char* stack;
};

struct Worker {
    Fiber fibers[MAX_THREADS];
    Fiber* cur = nullptr;
    void* sched_sp = nullptr;
    dim3 bidx, bdim, gdim;
    char* dyn_smem = nullptr;
    const std::function<void()>* body = nullptr;
    uint64_t xchg[MAX_WAVES][2][WAVE];
    float mfma_a[MAX_WAVES][2][WAVE];
    float mfma_b[MAX_WAVES][2][WAVE];
    unsigned mfma_a16[MAX_WAVES][2][WAVE][4];
    unsigned mfma_b16[MAX_WAVES][2][WAVE][4];
    bool reverse = false;

    Worker() {
        for (int i = 0; i < MAX_THREADS; ++i) {
            fibers[i].stack = (char*)mmap(nullptr, STACK_BYTES, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
            if (fibers[i].stack == (char*)MAP_FAILED) { perror("hipemu mmap"); abort(); }
        }
        void* p = nullptr;
        if (posix_memalign(&p, 256, DYN_SMEM_BYTES)) abort();
        dyn_smem = (char*)p;
        const char* r = getenv("HIPEMU_REVERSE");
        reverse = r && r[0] == '1';
    }
    ~Worker() {
        for (int i = 0; i < MAX_THREADS; ++i) munmap(fibers[i].stack, STACK_BYTES);
        free(dyn_smem);
    }
};

inline thread_local Worker* tl_worker = nullptr;

static void fiber_entry();

inline void yield_to_scheduler(int new_state) {
    Worker* w = tl_worker;
    Fiber* f = w->cur;
    f->state = new_state;
    hipemu_switch(&f->sp, w->sched_sp);
}

static void fiber_entry() {
    Worker* w = tl_worker;
    (*w->body)();
    yield_to_scheduler(DONE);
    abort();  // never resumed
}

inline void run_block(Worker* w, int nthreads) {
    const dim3 bd = w->bdim;
    for (int i = 0; i < nthreads; ++i) {
        Fiber& f = w->fibers[i];
        f.lin = i;
        f.tid = dim3(i % bd.x, (i / bd.x) % bd.y, i / (bd.x * bd.y));
        f.state = READY;
        f.wave_ops = 0;
        uintptr_t top = ((uintptr_t)f.stack + STACK_BYTES) & ~uintptr_t(15);
        void** sp = (void**)top;
        *--sp = nullptr;                 // fake return address slot (keeps rsp = 8 mod 16 at entry)
        *--sp = (void*)&fiber_entry;     // 'ret' target
        for (int r = 0; r < 6; ++r) *--sp = nullptr;  // rbp rbx r12 r13 r14 r15
        f.sp = sp;
    }
    int remaining = nthreads;
    const int nwaves = (nthreads + WAVE - 1) / WAVE;
    while (remaining > 0) {
        bool progressed = false;
        for (int k = 0; k < nthreads; ++k) {
            int i = w->reverse ? nthreads - 1 - k : k;
            Fiber& f = w->fibers[i];
            if (f.state != READY) continue;
            w->cur = &f;
            hipemu_switch(&w->sched_sp, f.sp);
            progressed = true;
            if (f.state == DONE) --remaining;
        }
        if (remaining == 0) break;
        // workgroup barrier: every live work-item has arrived
        int at_block = 0;
        for (int i = 0; i < nthreads; ++i) at_block += (w->fibers[i].state == WAIT_BLOCK);
        bool released = false;
        if (at_block == remaining) {
            for (int i = 0; i < nthreads; ++i)
                if (w->fibers[i].state == WAIT_BLOCK) w->fibers[i].state = READY;
            released = true;
        }
        // wave rendezvous: every live lane of the wave has arrived
        for (int wv = 0; wv < nwaves; ++wv) {
            int lo = wv * WAVE, hi = std::min(nthreads, lo + WAVE);
            int waiting = 0, live = 0;
            for (int i = lo; i < hi; ++i) {
                int s = w->fibers[i].state;
                live += (s != DONE);
                waiting += (s == WAIT_WAVE);
            }
            if (waiting > 0 && waiting == live) {
                for (int i = lo; i < hi; ++i)
                    if (w->fibers[i].state == WAIT_WAVE) w->fibers[i].state = READY;
                released = true;
            }
        }
        if (!progressed && !released) {
            fprintf(stderr, "hipemu: deadlock in block (%u,%u,%u): %d live work-items, %d at __syncthreads "
                            "(divergent barrier or wave op?)\n", w->bidx.x, w->bidx.y, w->bidx.z, remaining, at_block);
            abort();
        }
    }
}

// ---- persistent pool: one OS thread per core, blocks claimed from an atomic counter
struct Pool {
    std::vector<std::thread> threads;
    std::mutex mu;
    std::condition_variable cv_job, cv_done;
    const std::function<void()>* body = nullptr;
    dim3 grid, block;
    std::atomic<long> next{0};
    long total = 0;
    int active = 0;
    unsigned long generation = 0;
    bool stop = false;

    Pool() {
        unsigned n = std::thread::hardware_concurrency();
        const char* e = getenv("HIPEMU_THREADS");
        if (e) n = (unsigned)atoi(e);
        if (n < 1) n = 1;
        if (n > 64) n = 64;
        for (unsigned i = 0; i < n; ++i) threads.emplace_back([this] { this->loop(); });
    }
    ~Pool() {
        {
            std::lock_guard<std::mutex> lk(mu);
            stop = true;
        }
        cv_job.notify_all();
        for (auto& t : threads) t.join();
    }
    void loop() {
        Worker* w = new Worker();
        tl_worker = w;
        unsigned long seen = 0;
        for (;;) {
            {
                std::unique_lock<std::mutex> lk(mu);
                cv_job.wait(lk, [&] { return stop || generation != seen; });
                if (stop) break;
                seen = generation;
            }
            w->body = body;
            w->gdim = grid;
            w->bdim = block;
            const int nthreads = (int)(block.x * block.y * block.z);
            for (;;) {
                long b = next.fetch_add(1);
                if (b >= total) break;
                w->bidx = dim3((unsigned)(b % grid.x), (unsigned)((b / grid.x) % grid.y), (unsigned)(b / ((long)grid.x * grid.y)));
                run_block(w, nthreads);
            }
            {
                std::lock_guard<std::mutex> lk(mu);
                if (--active == 0) cv_done.notify_all();
            }
        }
        delete w;
    }
    void run(const std::function<void()>& fn, dim3 g, dim3 b) {
        std::unique_lock<std::mutex> lk(mu);
        body = &fn;
        grid = g;
        block = b;
        total = (long)g.x * g.y * g.z;
        next = 0;
        active = (int)threads.size();
        ++generation;
        cv_job.notify_all();
        cv_done.wait(lk, [&] { return active == 0; });
    }
};

inline Pool& pool() {
    static Pool* p = new Pool();  // leaked on purpose: worker threads outlive static destruction order
    return *p;
}
inline std::mutex& launch_mutex() {
    static std::mutex m;
    return m;
}

template <typename K, typename... Args>
inline void launch(K kernel, dim3 grid, dim3 block, size_t shmem, Args... args) {
    const size_t nthreads = (size_t)block.x * block.y * block.z;
    if (nthreads == 0 || nthreads > MAX_THREADS || shmem > DYN_SMEM_BYTES) {
        fprintf(stderr, "hipemu: bad launch config threads=%zu shmem=%zu\n", nthreads, shmem);
        abort();
    }
    if ((size_t)grid.x * grid.y * grid.z == 0) return;
    std::function<void()> fn = [=]() { kernel(args...); };
    std::lock_guard<std::mutex> lk(launch_mutex());  // one launch at a time (stream order)
    pool().run(fn, grid, block);
}

// ---- rendezvous helpers used by the device intrinsics below
inline void wave_sync() { yield_to_scheduler(WAIT_WAVE); }

template <typename T>
inline T wave_exchange(T v, int src_lane) {
    static_assert(sizeof(T) <= 8, "shuffle payload too large");
    Worker* w = tl_worker;
    Fiber* f = w->cur;
    const int wave = f->lin / WAVE, lane = f->lin % WAVE;
    const int slot = (f->wave_ops++) & 1;
    uint64_t bits = 0;
    memcpy(&bits, &v, sizeof(T));
    w->xchg[wave][slot][lane] = bits;
    wave_sync();
    T out;
    uint64_t got = w->xchg[wave][slot][src_lane & (WAVE - 1)];
    memcpy(&out, &got, sizeof(T));
    return out;
}

}  // namespace hipemu

#define threadIdx (hipemu::tl_worker->cur->tid)
#define blockIdx (hipemu::tl_worker->bidx)
#define blockDim (hipemu::tl_worker->bdim)
#define gridDim (hipemu::tl_worker->gdim)
#define warpSize 64

static inline void __syncthreads() { hipemu::yield_to_scheduler(hipemu::WAIT_BLOCK); }
static inline int hipemu_lane() { return hipemu::tl_worker->cur->lin % hipemu::WAVE; }

template <typename T> static inline T __shfl(T v, int src, int width = 64) {
    int lane = hipemu_lane();
    int base = lane & ~(width - 1);
    return hipemu::wave_exchange(v, base + (src & (width - 1)));
}
template <typename T> static inline T __shfl_xor(T v, int mask, int width = 64) {
    int lane = hipemu_lane();
    int src = lane ^ mask;
    if ((src & ~(width - 1)) != (lane & ~(width - 1))) src = lane;
    return hipemu::wave_exchange(v, src);
}
template <typename T> static inline T __shfl_down(T v, unsigned delta, int width = 64) {
    int lane = hipemu_lane();
    int src = lane + (int)delta;
    if ((src & ~(width - 1)) != (lane & ~(width - 1))) src = lane;
    return hipemu::wave_exchange(v, src);
}
template <typename T> static inline T __shfl_up(T v, unsigned delta, int width = 64) {
    int lane = hipemu_lane();
    int src = lane - (int)delta;
    if (src < 0 || (src & ~(width - 1)) != (lane & ~(width - 1))) src = lane;
    return hipemu::wave_exchange(v, src);
}
static inline unsigned long long __ballot(int pred) {
    unsigned long long m = 0;
    // 64 one-bit exchanges would be slow; exchange the predicate once per source lane group instead
    int lane = hipemu_lane();
    (void)lane;
    for (int src = 0; src < 64; ++src) {
        int p = hipemu::wave_exchange<int>(pred ? 1 : 0, src);
        m |= (unsigned long long)(p & 1) << src;
    }
    return m;
}

// ---- atomics (blocks run concurrently on several OS threads)
static inline float atomicAdd(float* p, float v) {
    uint32_t* ip = (uint32_t*)p;
    uint32_t old = __atomic_load_n(ip, __ATOMIC_RELAXED), nw;
    float f;
    do {
        memcpy(&f, &old, 4);
        f += v;
        memcpy(&nw, &f, 4);
    } while (!__atomic_compare_exchange_n(ip, &old, nw, false, __ATOMIC_RELAXED, __ATOMIC_RELAXED));
    memcpy(&f, &old, 4);
    return f;
}
static inline int atomicAdd(int* p, int v) { return __atomic_fetch_add(p, v, __ATOMIC_RELAXED); }
static inline unsigned atomicAdd(unsigned* p, unsigned v) { return __atomic_fetch_add(p, v, __ATOMIC_RELAXED); }
static inline unsigned atomicMax(unsigned* p, unsigned v) {
    unsigned old = __atomic_load_n(p, __ATOMIC_RELAXED);
    while (old < v && !__atomic_compare_exchange_n(p, &old, v, false, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {}
    return old;
}
static inline int atomicMax(int* p, int v) {
    int old = __atomic_load_n(p, __ATOMIC_RELAXED);
    while (old < v && !__atomic_compare_exchange_n(p, &old, v, false, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {}
    return old;
}

// ---- device math spelled the HIP way
// (__expf/__logf collide with glibc internals: kernels use expf/logf)
static inline float __fdividef(float a, float b) { return a / b; }
static inline float rsqrtf(float x) { return 1.0f / sqrtf(x); }
static inline float __frcp_rn(float x) { return 1.0f / x; }
static inline unsigned __float_as_uint(float f) { unsigned u; memcpy(&u, &f, 4); return u; }
static inline float __uint_as_float(unsigned u) { float f; memcpy(&f, &u, 4); return f; }
static inline int __float_as_int(float f) { int u; memcpy(&u, &f, 4); return u; }
static inline float __int_as_float(int u) { float f; memcpy(&f, &u, 4); return f; }
static inline unsigned __umulhi(unsigned a, unsigned b) { return (unsigned)(((unsigned long long)a * b) >> 32); }
static inline float __fmaf_rn(float a, float b, float c) { return fmaf(a, b, c); }

// ---- MFMA f32 forms (cdna_hip_programming.md §3): exact f32, k-ordered fmaf chain.
typedef float hipemu_f32x16 __attribute__((vector_size(64)));
typedef float hipemu_f32x4 __attribute__((vector_size(16)));

// v_mfma_f32_32x32x2_f32: A[i=l&31][k=l>>5], B[k=l>>5][j=l&31];
// C/D: col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5).
static inline hipemu_f32x16 hipemu_mfma_32x32x2(float a, float b, hipemu_f32x16 c) {
    hipemu::Worker* w = hipemu::tl_worker;
    hipemu::Fiber* f = w->cur;
    const int wave = f->lin / 64, lane = f->lin % 64;
    const int slot = (f->wave_ops++) & 1;
    w->mfma_a[wave][slot][lane] = a;
    w->mfma_b[wave][slot][lane] = b;
    hipemu::wave_sync();
    const float* A = w->mfma_a[wave][slot];
    const float* B = w->mfma_b[wave][slot];
    const int col = lane & 31;
    for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        float d = c[r];
        d = fmaf(A[row], B[col], d);            // k = 0
        d = fmaf(A[row + 32], B[col + 32], d);  // k = 1
        c[r] = d;
    }
    return c;
}
// v_mfma_f32_16x16x4_f32: A[i=l&15][k=l>>4], B[k=l>>4][j=l&15]; C/D: col = lane&15, row = (lane>>4)*4 + reg.
static inline hipemu_f32x4 hipemu_mfma_16x16x4(float a, float b, hipemu_f32x4 c) {
    hipemu::Worker* w = hipemu::tl_worker;
    hipemu::Fiber* f = w->cur;
    const int wave = f->lin / 64, lane = f->lin % 64;
    const int slot = (f->wave_ops++) & 1;
    w->mfma_a[wave][slot][lane] = a;
    w->mfma_b[wave][slot][lane] = b;
    hipemu::wave_sync();
    const float* A = w->mfma_a[wave][slot];
    const float* B = w->mfma_b[wave][slot];
    const int col = lane & 15;
    for (int r = 0; r < 4; ++r) {
        const int row = (lane >> 4) * 4 + r;
        float d = c[r];
        for (int k = 0; k < 4; ++k) d = fmaf(A[row + 16 * k], B[col + 16 * k], d);
        c[r] = d;
    }
    return c;
}

// v_cvt_pk_bf16_f32: round to nearest even, `lo` in bits [15:0]
static inline unsigned hipemu_bf16_rne(float f) {
    unsigned u = __float_as_uint(f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (u >> 16) | 0x40u;  // NaN stays NaN
    u += 0x7fffu + ((u >> 16) & 1u);
    return u >> 16;
}
static inline unsigned hipemu_cvt_pk_bf16_f32(float lo, float hi) { return hipemu_bf16_rne(lo) | (hipemu_bf16_rne(hi) << 16); }
// v_mfma_f32_32x32x16_bf16: lane l holds row (A) / column (B) l & 31 and the eight k-slots of half l >> 5 (slot e in
// bits [16 (e & 1) ...] of register e >> 1); C/D as the f32 32x32 form.  Products are exact; the sixteen of them are
// summed in double and added to c with one rounding (the hardware's internal order is unspecified; tests are toleranced).
// IEEE half <-> float in software (the f16 MFMA and v_cvt_pkrtz_f16_f32 of MATH_F16X2): round toward zero, saturating
static inline unsigned hipemu_f32_to_f16_rtz(float f) {
    unsigned u;
    memcpy(&u, &f, 4);
    const unsigned sign = (u >> 16) & 0x8000u;
    const unsigned ax = u & 0x7fffffffu;
    if (ax > 0x7f800000u) return sign | 0x7e00u;                 // NaN
    if (ax >= 0x477fe000u) return sign | (ax == 0x7f800000u ? 0x7c00u : 0x7bffu);  // >= 65504: inf stays inf, the rest saturates
    const int e = (int)(ax >> 23) - 127;
    if (e >= -14) return sign | (unsigned)(((e + 15) << 10) | ((ax >> 13) & 0x3ffu));  // normal: drop 13 mantissa bits
    if (e < -24) return sign;                                    // below the smallest subnormal
    const unsigned mant = (ax & 0x7fffffu) | 0x800000u;          // subnormal: value = mant * 2^(e - 23), unit 2^-24
    return sign | (mant >> (-e - 1));
}
static inline float hipemu_f16_to_f32(unsigned h) {
    const unsigned sign = (h & 0x8000u) << 16, e = (h >> 10) & 0x1fu, m = h & 0x3ffu;
    unsigned u;
    if (e == 0) {
        const float v = (float)m * 5.9604644775390625e-08f;  // 2^-24
        memcpy(&u, &v, 4);
        u |= sign;
    } else if (e == 31) {
        u = sign | 0x7f800000u | (m << 13);
    } else {
        u = sign | ((e + 112) << 23) | (m << 13);
    }
    float f;
    memcpy(&f, &u, 4);
    return f;
}
static inline hipemu_f32x16 hipemu_mfma_32x32x16_f16(uint4 a, uint4 b, hipemu_f32x16 c) {
    hipemu::Worker* w = hipemu::tl_worker;
    hipemu::Fiber* f = w->cur;
    const int wave = f->lin / 64, lane = f->lin % 64;
    const int slot = (f->wave_ops++) & 1;
    const unsigned av[4] = {a.x, a.y, a.z, a.w}, bv[4] = {b.x, b.y, b.z, b.w};
    for (int j = 0; j < 4; ++j) {
        w->mfma_a16[wave][slot][lane][j] = av[j];
        w->mfma_b16[wave][slot][lane][j] = bv[j];
    }
    hipemu::wave_sync();
    const int col = lane & 31;
    for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        double d = 0.0;
        for (int h = 0; h < 2; ++h)
            for (int e = 0; e < 8; ++e) {
                const unsigned ua = w->mfma_a16[wave][slot][row + 32 * h][e >> 1], ub = w->mfma_b16[wave][slot][col + 32 * h][e >> 1];
                const float fa = hipemu_f16_to_f32((e & 1) ? (ua >> 16) : (ua & 0xffffu));
                const float fb = hipemu_f16_to_f32((e & 1) ? (ub >> 16) : (ub & 0xffffu));
                d += (double)fa * (double)fb;
            }
        c[r] = (float)((double)c[r] + d);
    }
    return c;
}

static inline hipemu_f32x16 hipemu_mfma_32x32x16_bf16(uint4 a, uint4 b, hipemu_f32x16 c) {
    hipemu::Worker* w = hipemu::tl_worker;
    hipemu::Fiber* f = w->cur;
    const int wave = f->lin / 64, lane = f->lin % 64;
    const int slot = (f->wave_ops++) & 1;
    const unsigned av[4] = {a.x, a.y, a.z, a.w}, bv[4] = {b.x, b.y, b.z, b.w};
    for (int j = 0; j < 4; ++j) {
        w->mfma_a16[wave][slot][lane][j] = av[j];
        w->mfma_b16[wave][slot][lane][j] = bv[j];
    }
    hipemu::wave_sync();
    const int col = lane & 31;
    for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        double d = 0.0;
        for (int h = 0; h < 2; ++h)
            for (int e = 0; e < 8; ++e) {
                const unsigned ua = w->mfma_a16[wave][slot][row + 32 * h][e >> 1], ub = w->mfma_b16[wave][slot][col + 32 * h][e >> 1];
                const float fa = __uint_as_float((e & 1) ? (ua & 0xffff0000u) : (ua << 16));
                const float fb = __uint_as_float((e & 1) ? (ub & 0xffff0000u) : (ub << 16));
                d += (double)fa * (double)fb;
            }
        c[r] = (float)((double)c[r] + d);
    }
    return c;
}

// v_mfma_f32_16x16x32_bf16: lane l holds row (A) / column (B) l & 15 and the eight k-slots 8 (l >> 4) .. + 7 (slot j in bits
// [16 (j & 1) ...] of register j >> 1); C/D: col = lane & 15, row = 4 (lane >> 4) + reg.  Exact products, summed in double.
static inline hipemu_f32x4 hipemu_mfma_16x16x32_bf16(uint4 a, uint4 b, hipemu_f32x4 c) {
    hipemu::Worker* w = hipemu::tl_worker;
    hipemu::Fiber* f = w->cur;
    const int wave = f->lin / 64, lane = f->lin % 64;
    const int slot = (f->wave_ops++) & 1;
    const unsigned av[4] = {a.x, a.y, a.z, a.w}, bv[4] = {b.x, b.y, b.z, b.w};
    for (int j = 0; j < 4; ++j) {
        w->mfma_a16[wave][slot][lane][j] = av[j];
        w->mfma_b16[wave][slot][lane][j] = bv[j];
    }
    hipemu::wave_sync();
    const int col = lane & 15;
    for (int r = 0; r < 4; ++r) {
        const int row = 4 * (lane >> 4) + r;
        double d = 0.0;
        for (int qq = 0; qq < 4; ++qq)
            for (int e = 0; e < 8; ++e) {
                const unsigned ua = w->mfma_a16[wave][slot][row + 16 * qq][e >> 1], ub = w->mfma_b16[wave][slot][col + 16 * qq][e >> 1];
                const float fa = __uint_as_float((e & 1) ? (ua & 0xffff0000u) : (ua << 16));
                const float fb = __uint_as_float((e & 1) ? (ub & 0xffff0000u) : (ub << 16));
                d += (double)fa * (double)fb;
            }
        c[r] = (float)((double)c[r] + d);
    }
    return c;
}

#ifdef HIPEMU_IMPLEMENTATION
// The context switch: callee-saved registers + stack pointer (System V x86-64).
__asm__(R"(
.text
.globl hipemu_switch
.type hipemu_switch,@function
hipemu_switch:
    pushq %rbp
    pushq %rbx
    pushq %r12
    pushq %r13
    pushq %r14
    pushq %r15
    movq %rsp, (%rdi)
    movq %rsi, %rsp
    popq %r15
    popq %r14
    popq %r13
    popq %r12
    popq %rbx
    popq %rbp
    ret
.size hipemu_switch,.-hipemu_switch
)");
#endif
