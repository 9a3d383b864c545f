// eagle_backend.h -- glue shared by the eight replacement translation units of the R package's src/ (this directory).
// Compiled only where R + Rcpp exist (not in the build container of this repository).
#ifndef EAGLE_BACKEND_H
#define EAGLE_BACKEND_H
#include <Rcpp.h>

#include "eagle_hip.h"

// One context per R session, opened on first use: the GPUs listed in EAGLE_HIP_DEVICES ("0,1,2,3": every call shards its
// markers over them), else the device named by EAGLE_HIP_DEVICE, else device 0.
inline eagle_ctx* eagle_backend_ctx() {
    static eagle_ctx* ctx = nullptr;
    if (!ctx) {
        ctx = eagle_open_env();
        if (!ctx) Rcpp::stop(std::string("Eagle HIP backend: ") + eagle_open_error());
    }
    return ctx;
}

// The reference passes R's `message` closure into C++ (calculateMMt_rcpp.cpp:22,36); the library reports through a
// C callback on the calling (main R) thread, which forwards to that closure.
struct EagleMessageScope {
    Rcpp::Function fn;
    explicit EagleMessageScope(Rcpp::Function f) : fn(f) { eagle_set_message_callback(eagle_backend_ctx(), &EagleMessageScope::call, this); }
    ~EagleMessageScope() { eagle_set_message_callback(eagle_backend_ctx(), nullptr, nullptr); }
    static void call(const char* text, void* self) { static_cast<EagleMessageScope*>(self)->fn(text); }
};

inline void eagle_check(int rc) {
    if (rc < 0) Rcpp::stop(eagle_last_error(eagle_backend_ctx()));  // -> R error, as BEGIN_RCPP/END_RCPP would
}
#endif
