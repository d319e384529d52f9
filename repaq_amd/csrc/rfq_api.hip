// rfq_api.hip — context management and header plumbing of the C-ABI (include/rfq_hip.h).
#include "rfq_ctx.h"
#include <cstring>
#include <algorithm>
#include <new>
#include <cstdlib>
#include <string>
#include <cerrno>

extern "C" const char* rfq_version(void) {
#ifdef RFQ_SIMT_EMULATION
    return "rfq_hip 0.1.0 simt-emulation (test build, not a product path)";
#else
    return "rfq_hip 0.1.0 gfx950";
#endif
}

// name = the environment variable's name; value NULL or "" = back to the default.  Unknown names AND unknown values are errors (a typo must not pass for a
// default - ADVICE r4: atoll took "foo" for 0 and a negative size for 2^64 - x): numbers are parsed with an end-pointer check and a range per switch.
static bool opt_num(const std::string& v, long long lo, long long hi, long long* out) {
    if (v.empty()) return false;
    errno = 0; char* end = nullptr; const long long x = strtoll(v.c_str(), &end, 10);
    if (errno || !end || *end || end == v.c_str() || x < lo || x > hi) return false;
    *out = x; return true;
}
extern "C" int rfq_set_option(rfq_ctx* c, const char* name, const char* value) {
    if (!c || !name) return RFQ_E_ARG;
    const std::string n = name, v = value ? value : ""; const bool set = !v.empty(); const RfqOpts d; long long num = 0;
#define RFQ_OPT_NUM(LO, HI, WHAT) if (set && !opt_num(v, LO, HI, &num)) return rfq_fail(c, RFQ_E_ARG, "%s is " WHAT " (got \"%s\")", name, v.c_str());
    if (n == "RFQ_GATHER") { if (set && v != "old" && v != "tile") return rfq_fail(c, RFQ_E_ARG, "RFQ_GATHER is old or tile"); c->opt.gather_old = v == "old"; }
    else if (n == "RFQ_QUAL") { if (set && v != "bytes" && v != "masks") return rfq_fail(c, RFQ_E_ARG, "RFQ_QUAL is bytes or masks"); c->opt.qual_bytes = v == "bytes"; }
    else if (n == "RFQ_CODER") { if (set && v != "list" && v != "mask") return rfq_fail(c, RFQ_E_ARG, "RFQ_CODER is list or mask");
            c->opt.coder = v == "list" ? 1 : (v == "mask" ? 2 : 0); }
    else if (n == "RFQ_INDEX") { if (set && v != "2pass" && v != "1pass") return rfq_fail(c, RFQ_E_ARG, "RFQ_INDEX is 2pass or 1pass"); c->opt.index_2pass = v == "2pass"; }
    else if (n == "RFQ_IDX_TILES") { if (set && v != "4" && v != "8" && v != "16") return rfq_fail(c, RFQ_E_ARG, "RFQ_IDX_TILES is 4, 8 or 16");
            c->opt.idx_tiles = set ? atoi(v.c_str()) : d.idx_tiles; }
    else if (n == "RFQ_STREAMS") { if (set && v != "1" && v != "2") return rfq_fail(c, RFQ_E_ARG, "RFQ_STREAMS is 1 or 2"); c->opt.one_stream = v == "1"; }
    else if (n == "RFQ_SLICE_BYTES") { RFQ_OPT_NUM(1024, 0xFFFFFFF0ll - 16, "a number of bytes in 1024 .. 2^32 - 32") c->opt.slice_bytes = set ? (size_t)num : 0; }
    else if (n == "RFQ_SLICE_BASES") { RFQ_OPT_NUM(1, 0xFFFFFFF0ll, "a number of bases in 1 .. 2^32 - 16") c->opt.slice_bases = set ? (uint64_t)num : 0; }
    else if (n == "RFQ_WALK") { if (set && v != "exact" && v != "guess") return rfq_fail(c, RFQ_E_ARG, "RFQ_WALK is guess or exact"); c->opt.walk_exact = v == "exact"; }
    else if (n == "RFQ_GW_SHIFT") { RFQ_OPT_NUM(4, 30, "4 .. 30") c->opt.gw_shift = set ? (int)num : d.gw_shift; }
    else if (n == "RFQ_MATERIALISE") { if (set && v != "0" && v != "1") return rfq_fail(c, RFQ_E_ARG, "RFQ_MATERIALISE is 0 or 1"); c->opt.materialise = v == "1"; }
    else if (n == "RFQ_TRACE") { if (set && v != "0" && v != "1") return rfq_fail(c, RFQ_E_ARG, "RFQ_TRACE is 0 or 1"); c->opt.trace = v == "1"; }
    else if (n == "RFQ_G2_PAD") { RFQ_OPT_NUM(0, 150000, "0 .. 150000 bytes of LDS") c->opt.g2_pad = set ? (uint32_t)num : 0u; }
    else if (n == "RFQ_SP_PAD") { RFQ_OPT_NUM(0, 150000, "0 .. 150000 bytes of LDS") c->opt.sp_pad = set ? (uint32_t)num : d.sp_pad; }
    else if (n == "RFQ_POS_SEG") { if (set && v != "1024" && v != "2048") return rfq_fail(c, RFQ_E_ARG, "RFQ_POS_SEG is 1024 or 2048"); c->opt.pos_seg = set ? atoi(v.c_str()) : 0; }
    else if (n == "RFQ_SPEC") { if (set && v != "0" && v != "1") return rfq_fail(c, RFQ_E_ARG, "RFQ_SPEC is 0 or 1"); c->opt.no_spec = v == "0"; }
    else return rfq_fail(c, RFQ_E_ARG, "unknown option %s", name);
#undef RFQ_OPT_NUM
    return RFQ_OK;
}
static const char* const RFQ_OPTION_NAMES[] = { "RFQ_GATHER", "RFQ_QUAL", "RFQ_CODER", "RFQ_INDEX", "RFQ_IDX_TILES", "RFQ_STREAMS", "RFQ_SLICE_BYTES", "RFQ_SLICE_BASES", "RFQ_WALK", "RFQ_GW_SHIFT", "RFQ_MATERIALISE", "RFQ_TRACE", "RFQ_G2_PAD", "RFQ_SP_PAD", "RFQ_POS_SEG", "RFQ_SPEC" };
extern "C" const char* rfq_option_name(int i) { return (i >= 0 && i < (int)(sizeof RFQ_OPTION_NAMES / sizeof RFQ_OPTION_NAMES[0])) ? RFQ_OPTION_NAMES[i] : nullptr; }
// the switch's current value in the form rfq_set_option takes ("" = its default): what a caller saves before it changes a switch for a while
extern "C" int rfq_get_option(const rfq_ctx* c, const char* name, char* out, size_t cap) {
    if (!c || !name || !out || !cap) return RFQ_E_ARG;
    const std::string n = name; const RfqOpts& o = c->opt; const RfqOpts d; std::string v;
    if (n == "RFQ_GATHER") v = o.gather_old ? "old" : "";
    else if (n == "RFQ_QUAL") v = o.qual_bytes ? "bytes" : "";
    else if (n == "RFQ_CODER") v = o.coder == 1 ? "list" : (o.coder == 2 ? "mask" : "");
    else if (n == "RFQ_INDEX") v = o.index_2pass ? "2pass" : "";
    else if (n == "RFQ_IDX_TILES") v = o.idx_tiles != d.idx_tiles ? std::to_string(o.idx_tiles) : "";
    else if (n == "RFQ_STREAMS") v = o.one_stream ? "1" : "";
    else if (n == "RFQ_SLICE_BYTES") v = o.slice_bytes ? std::to_string(o.slice_bytes) : "";
    else if (n == "RFQ_SLICE_BASES") v = o.slice_bases ? std::to_string(o.slice_bases) : "";
    else if (n == "RFQ_WALK") v = o.walk_exact ? "exact" : "";
    else if (n == "RFQ_GW_SHIFT") v = o.gw_shift != d.gw_shift ? std::to_string(o.gw_shift) : "";
    else if (n == "RFQ_MATERIALISE") v = o.materialise ? "1" : "";
    else if (n == "RFQ_TRACE") v = o.trace ? "1" : "";
    else if (n == "RFQ_G2_PAD") v = o.g2_pad ? std::to_string(o.g2_pad) : "";
    else if (n == "RFQ_SP_PAD") v = o.sp_pad != d.sp_pad ? std::to_string(o.sp_pad) : "";
    else if (n == "RFQ_POS_SEG") v = o.pos_seg ? std::to_string(o.pos_seg) : "";
    else if (n == "RFQ_SPEC") v = o.no_spec ? "0" : "";
    else return RFQ_E_ARG;
    if (v.size() + 1 > cap) return RFQ_E_NOSPACE;
    memcpy(out, v.c_str(), v.size() + 1);
    return RFQ_OK;
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
    { int cu = 0; if (hipDeviceGetAttribute(&cu, hipDeviceAttributeMultiprocessorCount, device_id) == hipSuccess && cu > 0) c->n_cu = (uint32_t)cu; }
    if (hipStreamCreate(&c->stream) != hipSuccess) { delete c; return RFQ_E_NO_DEVICE; }
    c->own_stream = true;
    memset(&c->h_hdr, 0, sizeof c->h_hdr);
    for (const char* nm : RFQ_OPTION_NAMES) { const char* v = getenv(nm); if (v && *v && rfq_set_option(c, nm, v) != RFQ_OK) fprintf(stderr, "rfq_hip: ignoring %s=%s (%s)\n", nm, v, c->err.c_str()); }
    c->err.clear();
    *out = c;
    return RFQ_OK;
}

extern "C" void rfq_destroy(rfq_ctx* c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    (void)hipDeviceSynchronize();
    for (auto& b : c->b) b.release();
    c->d_hdr.release(); c->d_status.release(); c->d_cmp.release(); c->out_img.release(); c->out_fq1.release(); c->out_fq2.release(); c->out_acc.release();
            c->out_acc1.release(); c->out_acc2.release();
    c->timer.destroy();
    if (c->copy) { (void)hipStreamDestroy(c->copy); for (auto& e : c->copy_ev) if (e) (void)hipEventDestroy(e); }
    if (c->own_stream && c->stream) (void)hipStreamDestroy(c->stream);
    if (c->pin) (void)hipHostFree(c->pin);
    if (c->pin_up) (void)hipHostFree(c->pin_up);
    if (c->ev_up) (void)hipEventDestroy(c->ev_up);
    if (c->aux) (void)hipStreamDestroy(c->aux);
    if (c->ev_fork) (void)hipEventDestroy(c->ev_fork);
    if (c->ev_join) (void)hipEventDestroy(c->ev_join);
    if (c->ev_mid) (void)hipEventDestroy(c->ev_mid);
    if (c->aux2) (void)hipStreamDestroy(c->aux2);
    if (c->ev_f) (void)hipEventDestroy(c->ev_f);
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
extern "C" void rfq_clear_header(rfq_ctx* c) { if (c) { c->have_hdr = false; c->hdr_on_device = false; c->dense_ok = false; c->e3_pieces_failed = false; c->mixed_lengths = false; memset(&c->h_hdr, 0, sizeof c->h_hdr); } }

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
extern "C" int rfq_copy_h2d_async(rfq_ctx* c, void* d, const void* h, size_t n, uint64_t* ticket) {
    if (!c) return RFQ_E_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    if (!c->copy) HIPCHK(c, hipStreamCreateWithFlags(&c->copy, hipStreamNonBlocking));
    const uint64_t t = c->copy_next; hipEvent_t& ev = c->copy_ev[t & 63u];
    if (t >= 64 && c->copy_done + 64 <= t) { HIPCHK(c, hipEventSynchronize(ev)); c->copy_done = t - 63; }   // the slot's previous copy (64 tickets back)
    if (!ev) HIPCHK(c, hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    if (n) HIPCHK(c, hipMemcpyAsync(d, h, n, hipMemcpyHostToDevice, c->copy));
    HIPCHK(c, hipEventRecord(ev, c->copy));
    c->copy_next = t + 1;
    if (ticket) *ticket = t;
    return RFQ_OK;
}
extern "C" int rfq_copy_done(rfq_ctx* c, uint64_t ticket) {
    if (!c || ticket >= c->copy_next) return RFQ_E_ARG;
    if (ticket < c->copy_done || ticket + 64 < c->copy_next) return 1;      // (older than the ring: finished long ago)
    const hipError_t e = hipEventQuery(c->copy_ev[ticket & 63u]);
    if (e == hipSuccess) return 1;
    if (e == hipErrorNotReady) { (void)hipGetLastError(); return 0; }
    return rfq_fail(c, RFQ_E_HIP, "hipEventQuery failed: %s", hipGetErrorString(e));
}
extern "C" int rfq_copy_sync(rfq_ctx* c) {
    if (!c) return RFQ_E_ARG;
    if (c->copy) { HIPCHK(c, hipSetDevice(c->device)); HIPCHK(c, hipStreamSynchronize(c->copy)); c->copy_done = c->copy_next; }
    return RFQ_OK;
}
extern "C" int rfq_copy_d2d(rfq_ctx* c, void* dst, const void* src, size_t n) {
    if (!c) return RFQ_E_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    if (n) { HIPCHK(c, hipMemcpyAsync(dst, src, n, hipMemcpyDeviceToDevice, c->stream)); HIPCHK(c, hipStreamSynchronize(c->stream)); }
    return RFQ_OK;
}
// dst on this context's device <- src on src_ctx's device (the chunk ranges a multi-GPU host queue deals out): a peer copy over xGMI,
// which the runtime stages through the host when the two devices have no peer path
extern "C" int rfq_copy_peer(rfq_ctx* c, void* dst, const rfq_ctx* src_ctx, const void* src, size_t n) {
    if (!c || !src_ctx) return RFQ_E_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    if (n) {
        if (src_ctx->device == c->device) HIPCHK(c, hipMemcpyAsync(dst, src, n, hipMemcpyDeviceToDevice, c->stream));
        else HIPCHK(c, hipMemcpyPeerAsync(dst, c->device, src, src_ctx->device, n, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
    }
    return RFQ_OK;
}
// First byte at which two device texts differ (n when they are equal).  --compare on the device: a decoded batch
// that is byte-identical to the same span of the FASTQ file has, read for read, equal name / sequence / strand / quality
// (Repaq::compare's four tests, src/repaq.cpp:85-108); only a differing batch is cut into records by the caller to word the message.
// Pure HBM stream: 2 x n bytes read, one atomicMin per thread that saw a difference.
__global__ __launch_bounds__(256) void k_first_diff(const uint8_t* __restrict__ a, const uint8_t* __restrict__ b, uint64_t n, unsigned long long* __restrict__ out) {
    const uint64_t tid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x, stride = (uint64_t)gridDim.x * blockDim.x;
    unsigned long long best = ~0ull;
    if (((((uintptr_t)a) | ((uintptr_t)b)) & 15) == 0) {
        const uint64_t n16 = n >> 4;
        const uint4* a4 = (const uint4*)a; const uint4* b4 = (const uint4*)b;
        for (uint64_t i = tid; i < n16; i += stride) {
            const uint4 x = a4[i], y = b4[i];
            const uint32_t d[4] = { x.x ^ y.x, x.y ^ y.y, x.z ^ y.z, x.w ^ y.w };
            if (d[0] | d[1] | d[2] | d[3]) {
                for (int w = 0; w < 4; w++) if (d[w]) { best = i * 16 + w * 4 + ((__ffs((int)d[w]) - 1) >> 3); break; }
                break;                                                        // ascending i: the first hit of this thread is its smallest
            }
        }
        if (tid == 0) for (uint64_t i = n16 << 4; i < n; i++) if (a[i] != b[i]) { if (i < best) best = i; break; }
    } else {
        for (uint64_t i = tid; i < n; i += stride) if (a[i] != b[i]) { best = i; break; }
    }
    if (best != ~0ull) atomicMin(out, best);
}
extern "C" int rfq_compare_bytes(rfq_ctx* c, const void* d_a, const void* d_b, size_t n, uint64_t* first_diff) {
    if (!c || !first_diff || (n && (!d_a || !d_b))) return RFQ_E_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    *first_diff = n;
    if (!n) return RFQ_OK;
    HIPCHK(c, c->d_cmp.ensure(8));
    unsigned long long init = ~0ull, got = 0;
    HIPCHK(c, hipMemcpyAsync(c->d_cmp.p, &init, 8, hipMemcpyHostToDevice, c->stream));
    const uint64_t items = (n >> 4) + 1;
    const unsigned blocks = (unsigned)std::min<uint64_t>((items + 255) / 256, 256u * 16u);
    hipLaunchKernelGGL(k_first_diff, dim3(blocks), dim3(256), 0, c->stream, (const uint8_t*)d_a, (const uint8_t*)d_b, (uint64_t)n, c->d_cmp.as<unsigned long long>());
    KCHK(c, "k_first_diff");
    HIPCHK(c, hipMemcpyAsync(&got, c->d_cmp.p, 8, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (got != ~0ull) *first_diff = (uint64_t)got;
    return RFQ_OK;
}
extern "C" int rfq_host_alloc(rfq_ctx* c, void** p, size_t n) {
    if (!c || !p) return RFQ_E_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipHostMalloc(p, n ? n : 256, 0));
    return RFQ_OK;
}
extern "C" int rfq_host_free(rfq_ctx* c, void* p) { if (!c) return RFQ_E_ARG; if (p) HIPCHK(c, hipHostFree(p)); return RFQ_OK; }
// page-lock memory the caller already owns (hipHostRegister): a driver that starts reading its input before the HIP runtime is up locks those buffers afterwards
extern "C" int rfq_host_register(rfq_ctx* c, void* h_ptr, size_t n) {
    if (!c || !h_ptr || !n) return RFQ_E_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipHostRegister(h_ptr, n, hipHostRegisterDefault));
    return RFQ_OK;
}
extern "C" int rfq_host_unregister(rfq_ctx* c, void* h_ptr) {
    if (!c || !h_ptr) return RFQ_E_ARG;
    HIPCHK(c, hipHostUnregister(h_ptr));
    return RFQ_OK;
}
extern "C" int rfq_copy_d2h(rfq_ctx* c, void* h, const void* d, size_t n) {
    if (!c) return RFQ_E_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    if (n) { HIPCHK(c, hipMemcpyAsync(h, d, n, hipMemcpyDeviceToHost, c->stream)); HIPCHK(c, hipStreamSynchronize(c->stream)); }
    return RFQ_OK;
}

// ---------------------------------------------------------------- self test of the wave primitives (rfq_common.h)
// Every kernel's scans and reductions go through wave_incl_sum / wave_incl_max / wave_sum / ... - DPP row shifts and row broadcasts on the GPU,
// shuffles under the SIMT interpreter, which therefore cannot vouch for the DPP forms.  This entry point runs them on caller-chosen lane values so
// that a GPU test can sweep patterns across all 64 lanes against a serial reference (tests/test_wave_primitives.py).  out: 12 u64 per lane.
__global__ void k_selftest_wave(const unsigned long long* __restrict__ in, unsigned long long* __restrict__ out) {
    const uint32_t i = blockIdx.x * 64u + threadIdx.x; const unsigned long long v = in[i];
    const uint32_t a = (uint32_t)v; const int sa = (int)a;
    unsigned long long* o = out + 12ull * i;
    o[0] = wave_incl_sum<uint32_t>(a);
    o[1] = wave_incl_sum<unsigned long long>(v);
    o[2] = (unsigned long long)(long long)wave_incl_max<int>(sa);
    o[3] = (unsigned long long)wave_incl_max<long long>((long long)v);
    o[4] = wave_sum<uint32_t>(a);
    o[5] = wave_min<uint32_t>(a);
    o[6] = wave_max<uint32_t>(a);
    o[7] = ((unsigned long long)wave_and(a) << 32) | wave_or(a);
    o[8] = wave_shr1<uint32_t>(a, 0xABCD1234u);
    o[9] = wave_last<uint32_t>(a);
    o[10] = wave_min<unsigned long long>(v);
    { U4 u; u.a = a; u.b = a >> 3; u.c = a ^ 0x5A5Au; u.d = (uint32_t)(v >> 32); const U4 r = wave_incl_sum(u);
            o[11] = ((unsigned long long)(r.a + r.b + r.c) << 32) | r.d; }
}
extern "C" int rfq_selftest_wave(rfq_ctx* c, const uint64_t* h_in, uint32_t n_waves, uint64_t* h_out) {
    if (!c || !h_in || !h_out || !n_waves) return RFQ_E_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    const size_t n = (size_t)n_waves * 64u;
    void *din = nullptr, *dout = nullptr;
    HIPCHK(c, hipMalloc(&din, n * 8)); if (hipMalloc(&dout, n * 96) != hipSuccess) { (void)hipFree(din); return rfq_fail(c, RFQ_E_HIP, "hipMalloc failed"); }
    hipError_t e = hipMemcpy(din, h_in, n * 8, hipMemcpyHostToDevice);
    if (e == hipSuccess) { hipLaunchKernelGGL(k_selftest_wave, dim3(n_waves), dim3(64), 0, c->stream, (const unsigned long long*)din, (unsigned long long*)dout);
            e = hipGetLastError(); }
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    if (e == hipSuccess) e = hipMemcpy(h_out, dout, n * 96, hipMemcpyDeviceToHost);
    (void)hipFree(din); (void)hipFree(dout);
    if (e != hipSuccess) return rfq_fail(c, RFQ_E_HIP, "rfq_selftest_wave: %s", hipGetErrorString(e));
    return RFQ_OK;
}
