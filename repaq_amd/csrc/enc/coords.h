// enc/coords.h - coordinate coder (encodeCoords)
// Part of rfq_encode_kernels.h (included from there, in order; not a stand-alone header).
#pragma once
// =============================================================== coordinate coder (encodeCoords, src/rfqcodec.cpp:1262-1330)
// One wave per (axis, chunk).  `last` always equals the previous element, so every token is local: a repeat element
// closes a 0xC0|k token when it is the 32nd of its group or the next element is not a repeat.
__global__ void k_coords(ReadTab R, ChunkTab C, const DevHeader* __restrict__ D, uint8_t* __restrict__ xs, uint8_t* __restrict__ ys, DevStatus* st) {
    const uint32_t axis = blockIdx.x, c = blockIdx.y;
    if (!(D->flags & (axis ? H_Y : H_X))) return;
    const uint32_t f = C.first[c], e = C.first[c + 1]; const bool il = C.il[c] != 0;
    const uint32_t stride = il ? 2u : 1u, num = (e - f) / stride;
    const uint32_t* V = (axis ? R.y : R.x) + f;
    uint8_t* out = (axis ? ys : xs) + 3ull * f;
    const int l = lane_id();
    uint32_t outpos = 0, carry_prev = 1000u, carry_rep = 0; long long carry_start = -1;
    // (a step's values are requested two steps before they are coded: the steps are one dependent chain - carries, output position - and a load in it costs the
    // chain a round trip to memory per 64 values: 145 us for a chunk of 6,700 reads, alone at the end of the phase on a small input)
    auto fetch = [&](uint32_t base_) -> uint32_t { const uint32_t i_ = base_ + (uint32_t)l; return i_ < num ? V[(size_t)i_ * stride] : 0u; };
    uint32_t v1 = fetch(0u), v2 = fetch(64u);
    for (uint32_t base = 0; base < num; base += 64) {
        const uint32_t i = base + (uint32_t)l; const bool valid = i < num;
        const uint32_t v = v1; v1 = v2; v2 = fetch(base + 128u);
        const uint32_t p = wave_shr1(v, carry_prev);
        const uint32_t rep = (valid && v == p) ? 1u : 0u;
        // the element behind mine: the next lane's, for the last lane the next step's first
        uint32_t vnx = (uint32_t)__shfl_down((int)v, 1u); { const uint32_t first_next = wave_read(v1, 0u); if (l == 63) vnx = first_next; }
        const bool rep_next = (i + 1 < num) && vnx == v;
        const uint32_t rep_prev = wave_shr1(rep, carry_rep);
        long long sidx = (rep && !rep_prev) ? (long long)i : -1;
        sidx = wave_incl_max(sidx); if (carry_start > sidx) sidx = carry_start;
        uint32_t bytes = 0, kind = 0;                                       // kind 1: repeat close, 2: +diff, 3: 15-bit, 4: 21-bit
        if (valid) {
            if (rep) { const uint32_t k = (uint32_t)((long long)i - sidx); if (((k + 1) & 31u) == 0 || !rep_next) { bytes = 1; kind = 1; } }
            else {
                const int diff = (int)(v - p);
                if (diff > 0 && diff <= 64) { bytes = 1; kind = 2; }
                else if (v <= 32767u) { bytes = 2; kind = 3; }
                else if (v < (1u << 21)) { bytes = 3; kind = 4; }
                else { atomicOr(&st->err, (uint32_t)DE_COORD_RANGE);
                        atomicMin((unsigned long long*)&st->coord_key, ((unsigned long long)c << 34) | ((unsigned long long)axis << 33) | (unsigned long long)i); }
            }
        }
        const uint32_t incl = wave_incl_sum(bytes); uint32_t o = outpos + incl - bytes;
        if (kind == 1) out[o] = (uint8_t)(0xC0u | (((uint32_t)((long long)i - sidx)) & 31u));
        else if (kind == 2) out[o] = (uint8_t)(0x80u | (uint32_t)((int)(v - p) - 1));
        else if (kind == 3) { out[o] = (uint8_t)(v >> 8); out[o + 1] = (uint8_t)v; }
        else if (kind == 4) { out[o] = (uint8_t)((v >> 16) | 0xE0u); out[o + 1] = (uint8_t)(v >> 8); out[o + 2] = (uint8_t)v; }
        outpos += wave_last(incl);
        carry_prev = wave_last(v); carry_rep = wave_last(rep); carry_start = wave_last(sidx);
    }
    if (l == 0) { if (axis) C.ysize[c] = outpos; else C.xsize[c] = outpos; }
}
