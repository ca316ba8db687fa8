// c_api.cpp — extern "C" surface of libmi355vits.so (include/mi355vits.h).  Exceptions stop here:
// every entry point returns a status code and records a message, never aborts the process.
#include "c_api_internal.h"

namespace {
int create_common(WeightsFile& wf, int device, mi355vits_handle* out) {
    std::unique_ptr<mi355vits_engine> h(new mi355vits_engine());
    h->eng.reset(new Engine(wf, device));  // may throw: `h` is released by the unique_ptr
    *out = h.release();
    return MI355VITS_OK;
}
}  // namespace

extern "C" {

const char* mi355vits_version(void) {
#ifdef MI355_EMU
    return "mi355vits 0.1.0 (hipemu CPU test build)";
#else
    return "mi355vits 0.1.0 (gfx950)";
#endif
}

int mi355vits_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n < 0) return 0;
    return n;
}

int mi355vits_create(const char* weights_path, int device, mi355vits_handle* out) {
    if (out) *out = nullptr;
    return guarded(nullptr, [&] {
        if (!weights_path || !out) throw EngineError(MI355VITS_ERR_INVALID, "weights_path and out must not be null");
        WeightsFile wf;
        wf.load(weights_path);
        create_common(wf, device, out);
    });
}

int mi355vits_create_from_buffer(const void* blob, size_t blob_bytes, int device, mi355vits_handle* out) {
    if (out) *out = nullptr;
    return guarded(nullptr, [&] {
        if (!blob || !out) throw EngineError(MI355VITS_ERR_INVALID, "blob and out must not be null");
        WeightsFile wf;
        // copy into 64-byte aligned storage so tensor data is float-aligned whatever the caller passed
        wf.storage.resize(blob_bytes + 64);
        unsigned char* base = wf.storage.data();
        base += (64 - reinterpret_cast<uintptr_t>(base) % 64) % 64;
        memcpy(base, blob, blob_bytes);
        wf.parse(base, blob_bytes);
        create_common(wf, device, out);
    });
}

int mi355vits_clone(mi355vits_handle src, mi355vits_handle* out) {
    if (out) *out = nullptr;
    if (!src) return MI355VITS_ERR_INVALID;
    std::lock_guard<std::mutex> lk(src->eng->mu);  // the clone reads the source's math mode / flags: not while a run or set_math is under way
    return guarded(src, [&] {
        if (!out) throw EngineError(MI355VITS_ERR_INVALID, "out must not be null");
        std::unique_ptr<mi355vits_engine> h(new mi355vits_engine());
        h->eng.reset(new Engine(*src->eng));
        *out = h.release();
    });
}

int mi355vits_device_result(mi355vits_handle h, const int16_t** pcm, const float** audio, int64_t* row_stride,
                            int32_t* batch, const int32_t** device_lengths) {
    if (!h) return MI355VITS_ERR_INVALID;
    std::lock_guard<std::mutex> lk(h->eng->mu);
    return guarded(h, [&] {
        long rs = 0;
        int b = 0;
        h->eng->device_buffers(pcm, audio, &rs, &b, device_lengths);
        if (row_stride) *row_stride = rs;
        if (batch) *batch = b;
    });
}

void mi355vits_destroy(mi355vits_handle h) {
    if (!h) return;
    try {
        std::lock_guard<std::mutex> lk(h->eng->mu);
    } catch (...) {
    }
    delete h;
}

int mi355vits_set_math(mi355vits_handle h, int mode) {
    if (!h) return MI355VITS_ERR_INVALID;
    std::lock_guard<std::mutex> lk(h->eng->mu);
    return guarded(h, [&] { h->eng->set_math(mode); });
}
int mi355vits_get_math(mi355vits_handle h) {
    if (!h) return MI355VITS_ERR_INVALID;
    std::lock_guard<std::mutex> lk(h->eng->mu);
    return h->eng->math();
}

int mi355vits_get_config(mi355vits_handle h, mi355vits_config* out) {
    if (!h) return MI355VITS_ERR_INVALID;
    return guarded(h, [&] {
        if (!out) throw EngineError(MI355VITS_ERR_INVALID, "out must not be null");
        *out = h->eng->config();
    });
}

int mi355vits_run(mi355vits_handle h, const mi355vits_run_args* args, mi355vits_result* out) {
    if (out) memset(out, 0, sizeof(*out));  // before anything can fail: free_result below must never see garbage
    if (!h) return MI355VITS_ERR_INVALID;
    std::lock_guard<std::mutex> lk(h->eng->mu);
    int rc = guarded(h, [&] {
        if (!args) throw EngineError(MI355VITS_ERR_INVALID, "args must not be null");
        h->eng->run(*args, out);
    });
    if (rc != MI355VITS_OK && out) mi355vits_free_result(out);  // never hand back partial audio
    return rc;
}

int mi355vits_fetch(mi355vits_handle h, uint32_t want_flags, mi355vits_result* out) {
    if (out) memset(out, 0, sizeof(*out));
    if (!h) return MI355VITS_ERR_INVALID;
    std::lock_guard<std::mutex> lk(h->eng->mu);
    int rc = guarded(h, [&] { h->eng->fetch(want_flags, out); });
    if (rc != MI355VITS_OK && out) mi355vits_free_result(out);
    return rc;
}

void mi355vits_free_result(mi355vits_result* r) { free_result_impl(r); }

const char* mi355vits_last_error(mi355vits_handle h) { return h ? h->err.c_str() : create_error().c_str(); }

int mi355vits_profile_enable(mi355vits_handle h, int on) {
    if (!h) return MI355VITS_ERR_INVALID;
    std::lock_guard<std::mutex> lk(h->eng->mu);
    h->eng->profiler().enabled = on != 0;
    return MI355VITS_OK;
}
int mi355vits_profile_reset(mi355vits_handle h) {
    if (!h) return MI355VITS_ERR_INVALID;
    std::lock_guard<std::mutex> lk(h->eng->mu);
    return guarded(h, [&] { h->eng->profiler().clear(); });
}
long mi355vits_profile_report(mi355vits_handle h, char* buf, size_t cap) {
    if (!h || !buf || cap == 0) return MI355VITS_ERR_INVALID;
    std::lock_guard<std::mutex> lk(h->eng->mu);
    long n = 0;
    int rc = guarded(h, [&] {
        const std::string s = h->eng->profiler().report();
        n = (long)std::min(s.size(), cap - 1);
        memcpy(buf, s.data(), (size_t)n);
        buf[n] = 0;
    });
    return rc == MI355VITS_OK ? n : rc;
}

float mi355vits_last_run_ms(mi355vits_handle h) {
    if (!h) return -1.0f;
    std::lock_guard<std::mutex> lk(h->eng->mu);
    return h->eng->last_run_ms();
}

long mi355vits_get_tap(mi355vits_handle h, const char* name, float* out, size_t capacity, int64_t dims[4]) {
    if (!h || !name || !dims) return MI355VITS_ERR_INVALID;
    std::lock_guard<std::mutex> lk(h->eng->mu);
    long n = 0;
    int rc = guarded(h, [&] { n = h->eng->get_tap(name, out, capacity, dims); });
    return rc == MI355VITS_OK ? n : rc;
}
long mi355vits_get_tap_rows(mi355vits_handle h, const char* name, long row0, long nrows, float* out, size_t capacity, int64_t dims[4]) {
    if (!h || !name || !dims) return MI355VITS_ERR_INVALID;
    std::lock_guard<std::mutex> lk(h->eng->mu);
    long n = 0;
    int rc = guarded(h, [&] { n = h->eng->get_tap(name, out, capacity, dims, row0, nrows < 0 ? 0 : nrows); });
    return rc == MI355VITS_OK ? n : rc;
}
long mi355vits_list_taps(mi355vits_handle h, char* buf, size_t cap) {
    if (!h || !buf || cap == 0) return MI355VITS_ERR_INVALID;
    std::lock_guard<std::mutex> lk(h->eng->mu);
    const std::string s = h->eng->list_taps();
    const size_t n = std::min(s.size(), cap - 1);
    memcpy(buf, s.data(), n);
    buf[n] = 0;
    return (long)n;
}

}  // extern "C"
