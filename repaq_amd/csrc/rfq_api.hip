// rfq_api.hip — context management and header plumbing of the C-ABI (include/rfq_hip.h).
#include "rfq_ctx.h"
#include <cstring>
#include <new>

extern "C" const char* rfq_version(void) {
#ifdef RFQ_SIMT_EMULATION
    return "rfq_hip 0.1.0 simt-emulation (test build, not a product path)";
#else
    return "rfq_hip 0.1.0 gfx950";
#endif
}

extern "C" int rfq_create(rfq_ctx** out, int device_id) {
    if (!out) return RFQ_E_ARG;
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return RFQ_E_NO_DEVICE;   // no CPU fallback, by design
    if (device_id < 0 || device_id >= n) return RFQ_E_NO_DEVICE;
    if (hipSetDevice(device_id) != hipSuccess) return RFQ_E_NO_DEVICE;
    rfq_ctx* c = new (std::nothrow) rfq_ctx();
    if (!c) return RFQ_E_HIP;
    c->device = device_id;
    if (hipStreamCreate(&c->stream) != hipSuccess) { delete c; return RFQ_E_NO_DEVICE; }
    c->own_stream = true;
    memset(&c->h_hdr, 0, sizeof c->h_hdr);
    *out = c;
    return RFQ_OK;
}

extern "C" void rfq_destroy(rfq_ctx* c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    (void)hipDeviceSynchronize();
    for (auto& b : c->b) b.release();
    c->d_hdr.release(); c->d_status.release(); c->out_img.release(); c->out_fq1.release(); c->out_fq2.release();
    c->timer.destroy();
    if (c->own_stream && c->stream) (void)hipStreamDestroy(c->stream);
    if (c->aux) (void)hipStreamDestroy(c->aux);
    if (c->ev_fork) (void)hipEventDestroy(c->ev_fork);
    if (c->ev_join) (void)hipEventDestroy(c->ev_join);
    if (c->ev_mid) (void)hipEventDestroy(c->ev_mid);
    delete c;
}

extern "C" const char* rfq_last_error(const rfq_ctx* c) { return c ? c->err.c_str() : "null context"; }

extern "C" int rfq_set_stream(rfq_ctx* c, void* s) {
    if (!c) return RFQ_E_ARG;
    if (s) {
        if (c->own_stream && c->stream) (void)hipStreamDestroy(c->stream);
        c->stream = (hipStream_t)s; c->own_stream = false;
    } else if (!c->own_stream) {
        if (hipStreamCreate(&c->stream) != hipSuccess) return rfq_fail(c, RFQ_E_HIP, "hipStreamCreate failed");
        c->own_stream = true;
    }
    return RFQ_OK;
}

int rfq_upload_header(rfq_ctx* c, const uint8_t* h, size_t n);   // rfq_encode.hip

extern "C" int rfq_set_header(rfq_ctx* c, const uint8_t* h, size_t n) {
    if (!c || !h) return RFQ_E_ARG;
    c->err.clear();
    return rfq_upload_header(c, h, n);
}
extern "C" int rfq_get_header(rfq_ctx* c, uint8_t* out, size_t* len) {
    if (!c || !out || !len) return RFQ_E_ARG;
    if (!c->have_hdr) return rfq_fail(c, RFQ_E_STATE, "no header has been made or set on this context");
    memcpy(out, c->h_hdr.bytes, c->h_hdr.len); *len = c->h_hdr.len;
    return RFQ_OK;
}
extern "C" void rfq_clear_header(rfq_ctx* c) { if (c) { c->have_hdr = false; memset(&c->h_hdr, 0, sizeof c->h_hdr); } }

extern "C" int rfq_last_timings(const rfq_ctx* c, const char** names, float* ms, int cap) {
    if (!c) return 0;
    int n = 0;
    for (size_t i = 0; i < c->timer.ms.size() && n < cap; i++, n++) { if (names) names[n] = c->timer.names[i]; if (ms) ms[n] = c->timer.ms[i]; }
    return n;
}

extern "C" int rfq_dev_malloc(rfq_ctx* c, void** p, size_t n) {
    if (!c || !p) return RFQ_E_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipMalloc(p, n ? n : 256));
    return RFQ_OK;
}
extern "C" int rfq_dev_free(rfq_ctx* c, void* p) { if (!c) return RFQ_E_ARG; if (p) HIPCHK(c, hipFree(p)); return RFQ_OK; }
extern "C" int rfq_copy_h2d(rfq_ctx* c, void* d, const void* h, size_t n) {
    if (!c) return RFQ_E_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    if (n) { HIPCHK(c, hipMemcpyAsync(d, h, n, hipMemcpyHostToDevice, c->stream)); HIPCHK(c, hipStreamSynchronize(c->stream)); }
    return RFQ_OK;
}
extern "C" int rfq_copy_d2d(rfq_ctx* c, void* dst, const void* src, size_t n) {
    if (!c) return RFQ_E_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    if (n) { HIPCHK(c, hipMemcpyAsync(dst, src, n, hipMemcpyDeviceToDevice, c->stream)); HIPCHK(c, hipStreamSynchronize(c->stream)); }
    return RFQ_OK;
}
extern "C" int rfq_host_alloc(rfq_ctx* c, void** p, size_t n) {
    if (!c || !p) return RFQ_E_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipHostMalloc(p, n ? n : 256, 0));
    return RFQ_OK;
}
extern "C" int rfq_host_free(rfq_ctx* c, void* p) { if (!c) return RFQ_E_ARG; if (p) HIPCHK(c, hipHostFree(p)); return RFQ_OK; }
extern "C" int rfq_copy_d2h(rfq_ctx* c, void* h, const void* d, size_t n) {
    if (!c) return RFQ_E_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    if (n) { HIPCHK(c, hipMemcpyAsync(h, d, n, hipMemcpyDeviceToHost, c->stream)); HIPCHK(c, hipStreamSynchronize(c->stream)); }
    return RFQ_OK;
}
