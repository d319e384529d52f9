// rfq_ctx.h — host-side context: growable device buffers, error plumbing, stage timers, scan launcher.
#pragma once
#include "rfq_common.h"
#include "../../include/rfq_hip.h"
#include <cstring>
#include <string>
#include <vector>
#include <cstdio>
#include <cstdarg>
#include <algorithm>

struct DBuf {
    void* p = nullptr; size_t cap = 0;
    hipError_t ensure(size_t n) {
        if (n <= cap) return hipSuccess;
        if (p) { (void)hipFree(p); p = nullptr; cap = 0; }
        size_t want = n + n / 4 + 4096;
        hipError_t e = hipMalloc(&p, want);
        if (e != hipSuccess) { p = nullptr; return e; }
        cap = want; return hipSuccess;
    }
    // grow, keeping the first `keep` bytes (device-to-device copy on stream s)
    hipError_t ensure_keep(size_t n, size_t keep, hipStream_t s) {
        if (n <= cap) return hipSuccess;
        void* q = nullptr; const size_t want = n + n / 2 + 4096;
        hipError_t e = hipMalloc(&q, want);
        if (e != hipSuccess) return e;
        if (p && keep) { e = hipMemcpyAsync(q, p, keep, hipMemcpyDeviceToDevice, s); if (e == hipSuccess) e = hipStreamSynchronize(s);
                if (e != hipSuccess) { (void)hipFree(q); return e; } }
        if (p) (void)hipFree(p);
        p = q; cap = want; return hipSuccess;
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }
    template <class T> T* as() const { return (T*)p; }
};

struct StageTimer {
    struct Stage { const char* name; hipEvent_t a, b; };
    std::vector<Stage> stages; size_t used = 0; std::vector<float> ms; std::vector<const char*> names;
    void reset() { used = 0; }
    void begin(const char* name, hipStream_t s) {
        if (used == stages.size()) { Stage st; st.name = name; (void)hipEventCreate(&st.a); (void)hipEventCreate(&st.b); stages.push_back(st); }
        stages[used].name = name; (void)hipEventRecord(stages[used].a, s);
    }
    void end(hipStream_t s) { (void)hipEventRecord(stages[used].b, s); used++; }
    void collect() {   // stream must be synchronised
        ms.assign(used, 0.f); names.assign(used, nullptr);
        for (size_t i = 0; i < used; i++) { (void)hipEventElapsedTime(&ms[i], stages[i].a, stages[i].b); names[i] = stages[i].name; }
    }
    void destroy() { for (auto& s : stages) { (void)hipEventDestroy(s.a); (void)hipEventDestroy(s.b); } stages.clear(); }
};

// Workgroups per chunk for a tile kernel whose grid is (x, n_chunks): the GPU holds `slots` of its workgroups at a time, and a grid that is not
// a whole number of such rounds leaves the last round part empty (3360 chunks x 2 = 6720 workgroups on 1280 slots are 5.25 rounds: measured
// 12 % slower than x 3 = 7.9 rounds).  Among the grids of about 6 .. 12 rounds the one that fills its last round best; ties go to the smaller.
static inline uint32_t grid_x_for(uint32_t n_chunks, uint32_t tiles_per_chunk, uint32_t slots) {
    uint32_t best = 1; double best_eff = 0.0;
    if (!n_chunks || !slots) return 1;
    for (uint32_t k = 6; k <= 12; k++) {
        uint32_t bx = (uint32_t)(((uint64_t)k * slots + n_chunks / 2) / n_chunks);
        bx = std::max<uint32_t>(1u, std::min<uint32_t>(std::max<uint32_t>(bx, 1u), tiles_per_chunk));
        const double rounds = (double)bx * n_chunks / slots, full = (double)(uint64_t)(rounds + 0.999999), eff = full > 0 ? rounds / full : 0.0;
        if (eff > best_eff + 1e-3) { best = bx; best_eff = eff; }
    }
    return best;
}
// Test / diagnostic switches of a context (rfq_set_option).  The RFQ_* environment variables of the same names are read ONCE, when the context is created
// (no getenv on the batch paths - the switches changed chunk walking and emission silently per call, and getenv races a concurrent setenv).
struct RfqOpts {
    // RFQ_CODER=list|mask   encode, quality bytes: the list coder / the mask coder whatever the number of coded values (default: list from five on)
    int  coder = 0;
    // RFQ_QUAL=bytes     encode: k_gather2 writes the quality bytes (qcat) also for files with <= 4 coded values (default there: match masks)
    bool qual_bytes = false;
    bool gather_old = false;          // RFQ_GATHER=old     encode: the byte-wise gather (k_gather + k_packbytes) also for reads that fit a tile
    bool index_2pass = false;         // RFQ_INDEX=2pass    encode: newline bitmap -> scan -> line offsets instead of the one-pass index
    int  idx_tiles = 0;               // RFQ_IDX_TILES=4|8|16   text per workgroup of the one-pass index (x 16 KiB); 0 = default
    bool one_stream = false;          // RFQ_STREAMS=1      no second stream: every kernel of a batch on the context's stream
    size_t slice_bytes = 0;           // RFQ_SLICE_BYTES    encode: slices of that many bytes per stream (so that the slicing logic runs on small inputs)
    uint64_t slice_bases = 0;         // RFQ_SLICE_BASES    decode: ranges of that many bases
    // RFQ_WALK=exact     decode without a chunk index: straight to the exact serial walk (default: guess and verify, that walk behind it)
    bool walk_exact = false;
    int  gw_shift = 16;               // RFQ_GW_SHIFT       log2 of the smallest guess-and-verify segment
    bool materialise = false;         // RFQ_MATERIALISE=1  decode: qualities / bases expanded in HBM (the path of a streaming caller's non-final slices) on every call
    bool trace = false;               // RFQ_TRACE          a line on stderr about how chunk starts were found
    uint32_t g2_pad = 0;              // RFQ_G2_PAD         profiling aid: bytes of unused dynamic LDS added to k_gather2 (fewer resident workgroups)
    // RFQ_SP_PAD         bytes of unused dynamic LDS added to k_seqpack: caps its resident workgroups so that the position coder beside it keeps its share
    uint32_t sp_pad = 0;
    bool no_spec = false;             // RFQ_SPEC=0         decode: the emitter only behind the host's look at the status (default: launched ahead of it where the caller gave the output buffers)
    int  pos_seg = 0;                 // RFQ_POS_SEG=1024|2048   decode, list chain: bytes of a position stream per wave (default: by the largest stream, dec/pos_lists.h)
};
struct rfq_ctx {
    RfqOpts opt;
    bool e3_pieces_failed = false;       // decode: a tile of k_dec_emit3 did not hold its reads' name pieces - files like this one go to the expanded path
    int device = 0; uint32_t n_cu = 256;                                         // compute units of the device (rfq_create)
    hipStream_t stream = nullptr; bool own_stream = false;
    // second stream for small latency-bound kernels that are independent of the main chain (coordinate coder / decoder): fork with
    // ev_fork recorded on `stream`, join with ev_join recorded on `aux`
    hipStream_t aux = nullptr; hipEvent_t ev_fork = nullptr, ev_join = nullptr, ev_mid = nullptr;
    hipStream_t aux2 = nullptr; hipEvent_t ev_f = nullptr;                     // decode: the bandwidth-bound prefill beside the latency-bound chains
    hipEvent_t ev_ovl = nullptr;                                               // encode: the overlap search (aux) has finished
    // rfq_copy_h2d_async: a stream of its own and a ring of events (ticket t lives in slot t % 64; a slot is reused only when its copy is done)
    hipStream_t copy = nullptr; hipEvent_t copy_ev[64] = {}; uint64_t copy_next = 0, copy_done = 0;
    bool aux_ready() {
        if (aux) return true;
        // (the second stream carries the chains of small, latency-bound kernels - stored prefix, sequence packer, N coder, coordinates; decode: the list
        // chain - beside one large VALU-bound kernel on the main stream.  A higher stream priority for it was measured in round 4: nothing on
        // encode, 0.3 ms worse on decode - plain streams)
        if (hipStreamCreateWithFlags(&aux, hipStreamNonBlocking) != hipSuccess) { aux = nullptr; return false; }
        if (hipEventCreateWithFlags(&ev_fork, hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&ev_join, hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&ev_mid, hipEventDisableTiming) != hipSuccess) return false;
        if (hipStreamCreateWithFlags(&aux2, hipStreamNonBlocking) != hipSuccess) { aux2 = nullptr; return false; }
        if (hipEventCreateWithFlags(&ev_f, hipEventDisableTiming) != hipSuccess) return false;
        if (hipEventCreateWithFlags(&ev_ovl, hipEventDisableTiming) != hipSuccess) return false;
        return true;
    }
    std::string err;
    // Small device -> host read-backs (status words, totals, chunk offsets) land in ONE page-locked block and are handed to their
    // destinations after the stream is synchronised: with a pageable destination every hipMemcpyAsync is a staged, effectively
    // synchronous copy (three in a row cost ~60 us of idle GPU between the decoder's kernels).
    uint8_t* pin = nullptr; size_t pin_cap = 0, pin_used = 0;
    struct Pend { void* host; size_t off, n; }; std::vector<Pend> pend;
    hipError_t fetch(void* host, const void* dev, size_t n, hipStream_t s) {
        if (!n) return hipSuccess;
        if (!pin) { if (hipHostMalloc((void**)&pin, 1 << 16, 0) != hipSuccess) { pin = nullptr; (void)hipGetLastError(); } else pin_cap = 1 << 16; }
        const size_t off = (pin_used + 15) & ~(size_t)15;
        if (!pin || off + n > pin_cap) return hipMemcpyAsync(host, dev, n, hipMemcpyDeviceToHost, s);     // too large for the block: straight to its destination
        pend.push_back(Pend{ host, off, n }); pin_used = off + n;
        return hipMemcpyAsync(pin + off, dev, n, hipMemcpyDeviceToHost, s);
    }
    hipError_t fetch_sync(hipStream_t s) {
        const hipError_t e = hipStreamSynchronize(s);
        if (e == hipSuccess) for (const Pend& q : pend) memcpy(q.host, pin + q.off, q.n);
        pend.clear(); pin_used = 0;
        return e;
    }
    // header
    DBuf d_hdr;                 // DevHeader
    DevHeader h_hdr; bool have_hdr = false;
    // rfq_upload_header: the device copy of h_hdr is current (a header MADE by the encoder lives on the device first and is fetched); the page-locked block uploads go through
    bool hdr_on_device = false; uint8_t* pin_up = nullptr; hipEvent_t ev_up = nullptr; bool ev_up_pending = false;
    DBuf d_status; DevStatus h_status;
    DBuf d_cmp;                 // rfq_compare_bytes: one u64 (first differing offset)
    // generic named buffers (see rfq_encode.hip / rfq_decode.hip)
    DBuf b[120];
    DBuf out_img, out_fq1, out_fq2, out_acc, out_acc1, out_acc2;       // out_acc*: the results of a sliced encode / decode call, appended
    // encode without a read-back behind the line index: records per byte of the last batch (0: not known yet - the first batch of a context reads back), and
    // "this call repeats a batch the lazy form could not take"
    double rec_per_byte = 0.0; bool lazy_block = false;
    bool retried_room = false;             // encode: the call in progress repeats a batch whose arenas were too small (a marker for rfq_last_timings)
    bool mixed_lengths = false;            // encode: this file has reads of several lengths (the prefix scans are launched up front; reset with the header)
    bool dense_ok = false;                 // encode, match-mask mode: DevHeader::dense of the device header is set (k_dense_order; reset with the header)
    // encode, match-mask mode: the rare planes of b[B_QPLANE] may hold bits (fresh buffer, or a call that left early): zero them whole
    bool qplane_dirty = true;
    uint32_t qplane_nd = 0, qplane_mask = 0; // ... dense planes the last batch used: how many, which (the others must be all-zero)
    size_t qplane_stride = 0;              // ... words per plane the buffer is laid out with (fixed while the buffer is)
    std::vector<uint64_t> chunk_off;
    std::vector<uint64_t> scan_end[2];     // rfq_scan_batch: end offset of every chunk in each input stream
    StageTimer timer;
};

static inline int rfq_fail(rfq_ctx* c, int code, const char* fmt, ...) {
    char buf[1024]; va_list ap; va_start(ap, fmt); vsnprintf(buf, sizeof buf, fmt, ap); va_end(ap);
    if (c) c->err = buf;
    return code;
}
#define HIPCHK(ctx, call) do { hipError_t e_ = (call); if (e_ != hipSuccess) return rfq_fail(ctx, RFQ_E_HIP, "%s failed: %s", #call, hipGetErrorString(e_)); } while (0)
#define KCHK(ctx, what) do { hipError_t e_ = hipGetLastError(); if (e_ != hipSuccess) return rfq_fail(ctx, RFQ_E_HIP, "launch of %s failed: %s", what, hipGetErrorString(e_)); } while (0)

// ---------------------------------------------------------------- multi-block exclusive scan (3 launches)
#define SCAN_TPB 256
#define SCAN_ITEMS 8
#define SCAN_TILE (SCAN_TPB * SCAN_ITEMS)

// skip (optional): a device word - non-zero means "the caller has a closed form for this prefix": the launch returns at once (k_lens_uniform, rfq_encode.hip)
template <class T> __global__ void k_scan_reduce(const T* __restrict__ in, T* __restrict__ partial, uint64_t n, const uint32_t* __restrict__ skip) {
    if (skip && *skip) return;
    // (a sum does not care about order: item i of thread t is element i * SCAN_TPB + t of the tile, so every load is coalesced)
    const uint64_t base = (uint64_t)blockIdx.x * SCAN_TILE + (uint64_t)threadIdx.x;
    T v[SCAN_ITEMS];
#pragma unroll
    for (int i = 0; i < SCAN_ITEMS; i++) v[i] = base + (uint64_t)i * SCAN_TPB < n ? in[base + (uint64_t)i * SCAN_TPB] : T();
    T acc = T();
#pragma unroll
    for (int i = 0; i < SCAN_ITEMS; i++) acc = acc + v[i];
    T tot; (void)block_excl_sum<T>(acc, &tot);
    if (threadIdx.x == 0) partial[blockIdx.x] = tot;
}
// the tiles' sums -> their exclusive prefixes, by ONE workgroup: every thread takes a run of consecutive partials (summed, then re-walked with
// its offset), one block scan in between (the earlier form - a block scan and two barriers per 256 partials - took 48 us for 11 k tiles)
template <class T> __global__ void k_scan_partials(T* __restrict__ partial, uint32_t nb, T* __restrict__ grand, const uint32_t* __restrict__ skip) {
    if (skip && *skip) return;
    const uint32_t K = (nb + SCAN_TPB - 1) / SCAN_TPB, i0 = threadIdx.x * K, i1 = i0 + K < nb ? i0 + K : nb;
    T acc = T();
    for (uint32_t i = i0; i < i1; i++) acc = acc + partial[i];
    T tot; T run = block_excl_sum<T>(acc, &tot);
    for (uint32_t i = i0; i < i1; i++) { const T v = partial[i]; partial[i] = run; run = run + v; }
    if (threadIdx.x == 0 && grand) *grand = tot;
}
// out[i] = exclusive prefix; out may alias in.  When out has n+1 entries pass write_total=1 to store the total at out[n].
template <class T> __global__ void k_scan_apply(const T* __restrict__ in, T* __restrict__ out, const T* __restrict__ partial, uint64_t n, int write_total, const uint32_t* __restrict__ skip) {
    if (skip && *skip) return;
    const uint64_t base = (uint64_t)blockIdx.x * SCAN_TILE + (uint64_t)threadIdx.x * SCAN_ITEMS;
    T v[SCAN_ITEMS]; T acc = T();
    for (int i = 0; i < SCAN_ITEMS; i++) { v[i] = base + i < n ? in[base + i] : T(); acc = acc + v[i]; }
    T tot; T ex = block_excl_sum<T>(acc, &tot);
    T run = partial[blockIdx.x] + ex;
    for (int i = 0; i < SCAN_ITEMS; i++) { if (base + i < n) out[base + i] = run; run = run + v[i]; }
    if (write_total && base < n && base + SCAN_ITEMS >= n) out[n] = run;
}
// a short array (per-chunk quantities: a few thousand elements) by ONE workgroup in ONE launch: every thread sums its run of consecutive elements, one block
// scan, the runs re-walked with their offsets (three launches took 30 us each way for 3,360 elements, and an encode step has five of these)
#define SCAN_SMALL 16384u
template <class T> __global__ void k_scan_small(const T* in, T* out, uint32_t n, int write_total, const uint32_t* __restrict__ skip) {
    if (skip && *skip) return;
    const uint32_t K = (n + SCAN_TPB - 1) / SCAN_TPB, i0 = threadIdx.x * K, i1 = i0 + K < n ? i0 + K : n;
    T acc = T();
    for (uint32_t i = i0; i < i1; i++) acc = acc + in[i];
    T tot; T run = block_excl_sum<T>(acc, &tot);
    for (uint32_t i = i0; i < i1; i++) { const T v = in[i]; out[i] = run; run = run + v; }     // (out may alias in: a thread reads an element before it writes it)
    if (write_total && threadIdx.x == 0) out[n] = tot;
}
// tmp must hold ceil(n / SCAN_TILE) + 1 elements of T
template <class T> static inline void scan_exclusive(hipStream_t s, const T* in, T* out, uint64_t n, T* tmp, int write_total, const uint32_t* skip = nullptr) {
    if (n == 0) { if (write_total) (void)hipMemsetAsync(out, 0, sizeof(T), s); return; }
    if (n <= SCAN_SMALL) { hipLaunchKernelGGL((k_scan_small<T>), dim3(1), dim3(SCAN_TPB), 0, s, in, out, (uint32_t)n, write_total, skip); return; }
    const uint32_t nb = (uint32_t)((n + SCAN_TILE - 1) / SCAN_TILE);
    hipLaunchKernelGGL((k_scan_reduce<T>), dim3(nb), dim3(SCAN_TPB), 0, s, in, tmp, n, skip);
    hipLaunchKernelGGL((k_scan_partials<T>), dim3(1), dim3(SCAN_TPB), 0, s, tmp, nb, (T*)nullptr, skip);
    hipLaunchKernelGGL((k_scan_apply<T>), dim3(nb), dim3(SCAN_TPB), 0, s, in, out, (const T*)tmp, n, write_total, skip);
}

// ---------------------------------------------------------------- several fills in one launch
// The per-batch tables that start all-zero / all-ones, in ONE launch (they were up to eight hipMemsetAsync calls - a fill kernel each, ~5 us on the device and ~10 us of
// host enqueue apiece, in chains where the GPU waits for the host: profiles/r06_b_queue_timeline_256.txt).  words: 32-bit words of each region; val: the word it is filled with.
#define CLEAR_MAX 8
struct ClearList { uint32_t* p[CLEAR_MAX]; uint64_t words[CLEAR_MAX]; uint32_t val[CLEAR_MAX]; uint32_t n;
    void add(void* p_, size_t bytes, uint32_t v) { p[n] = (uint32_t*)p_; words[n] = (bytes + 3) / 4; val[n] = v; n++; } };
template <int UNUSED> __global__ void k_clear_list(ClearList z) {
    const uint64_t t0 = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x, NT = (uint64_t)gridDim.x * blockDim.x;
    for (uint32_t r = 0; r < z.n; r++) {                                    // (uniform)
        uint32_t* const p = z.p[r]; const uint32_t v = z.val[r]; const uint64_t n4 = z.words[r] >> 2;
        for (uint64_t i = t0; i < n4; i += NT) ((uint4*)p)[i] = make_uint4(v, v, v, v);                       // (regions start on 16-byte boundaries)
        for (uint64_t i = (n4 << 2) + t0; i < z.words[r]; i += NT) p[i] = v;
    }
}
static inline void clear_list(hipStream_t s, const ClearList& z) {
    size_t words = 0; for (uint32_t i = 0; i < z.n; i++) words = std::max<size_t>(words, (size_t)z.words[i]);
    if (z.n) hipLaunchKernelGGL((k_clear_list<0>), dim3((uint32_t)std::min<size_t>(2048, words / 1024 + 1)), dim3(256), 0, s, z);
}
