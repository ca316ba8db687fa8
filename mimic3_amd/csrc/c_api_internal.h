// c_api_internal.h — what c_api.cpp (the product ABI, include/mi355vits.h) and lab_api.cpp (the test / bench / probe hooks,
// include/mi355vits_lab.h) share: the handle, the exception fence, a device buffer.
#pragma once
#include <cstring>
#include <memory>
#include <string>
#include <vector>

#include "engine.h"

using namespace m355;

struct mi355vits_engine {
    std::unique_ptr<Engine> eng;
    std::string err;
};

// message of a failed call that has no handle (create, the device-level hooks): one per thread and library, read by
// mi355vits_last_error(NULL) — external linkage so that both translation units see the same string
inline std::string& create_error() { static thread_local std::string e; return e; }

namespace {

template <typename F>
int guarded(mi355vits_handle h, F&& fn) {
    std::string* err = h ? &h->err : &create_error();
    try {
        fn();
        return MI355VITS_OK;
    } catch (const EngineError& e) {
        *err = e.what();
        return e.code;
    } catch (const std::bad_alloc&) {
        *err = "out of host memory";
        return MI355VITS_ERR_NOMEM;
    } catch (const std::exception& e) {
        *err = e.what();
        return std::string(e.what()).rfind("HIP error", 0) == 0 ? MI355VITS_ERR_DEVICE : MI355VITS_ERR_INTERNAL;
    } catch (...) {
        *err = "unknown error";
        return MI355VITS_ERR_INTERNAL;
    }
}

}  // namespace

namespace {
struct DevBuf {
    void* p = nullptr;
    explicit DevBuf(size_t bytes) { HIP_CHECK(hipMalloc(&p, bytes ? bytes : 16)); }
    ~DevBuf() { (void)hipFree(p); }
    template <typename T> T* as() { return static_cast<T*>(p); }
};
}  // namespace

