// conv_v7.h -- included by conv.hip INSIDE its anonymous namespace (shares ConvArgs, Mfma, epilogue_wave, ...).
//
// v7: persistent, stream-K, halo-patch implicit GEMM for the 3x3 / stride 1 / pad 1 convolutions with Cout % 256 == 0
// (reference models/common.py:57-81 Conv inside Bottleneck.cv2, models/yolov3.yaml:23-31 and the 3x3 convs of the head) --
// 24 of the 29 equal-FLOP 3x3 launches of a yolov3 forward, and the data gradients of the same layers.
//
// What v6 (conv_igemm_v6_kernel) left on the table at the BASELINE shapes (profiles/r01_*):
//   (1) 200 / 400 / 800 tiles of 256x256 on 256 CUs: every launch pays for whole CU-rounds it fills to 78 %;
//   (2) every one of the 9 taps re-stages the SAME input pixels (shifted by one row / column) from L2 into LDS: per 32-channel
//       block 9 x 16 KiB of activations + 9 x 16 KiB of filters = 288 `buffer_load ... lds` pieces, and the MEM phase of a
//       wave (4 pieces + their address arithmetic + 12 fragment reads) is longer than the 16 MFMAs it has to hide behind.
// v7:
//   * halo patch: for a 32-channel block the tile's input is staged ONCE as the flattened pixel range
//     [m0 - W - 1, m0 + 256 + W + 1) x 32 channels (XP x 8 KiB, XP = 3..5); tap (dh, dw) reads its B fragments at row offset
//     dh * W + dw of that patch.  Pixels whose tap falls outside the image (left / right / top / bottom edge -- in the flattened
//     index those neighbours are real pixels of the previous / next row or image) read a 64-byte zero block instead: one
//     v_cndmask on the LDS address per fragment, no data masking.  DMA pieces per block of 32 channels: 144 (filters) + 8 XP
//     instead of 288; per wave and K-step 2 filter pieces (+ 1 patch piece in XP of the 9 steps) instead of 4.
//   * persistent grid (one 8-wave block per CU) over the linearised (tile, channel block) space, split EVENLY (stream-K): a block
//     owns a contiguous range of units and walks it from the END.  A tile cut by a range boundary is finished by the block that
//     owns its LAST channel block: that block reaches the tile at the end of its walk, while the owners of the tile's head computed
//     their part FIRST and published it as an fp32 partial slab (256 KiB; agent-scope release / acquire, guide G16).  Block ids
//     come from an atomic ticket and a consumer only waits for EARLIER tickets of its own group, i.e. for blocks that are already
//     running and publish before anything else: no residency or dispatch-order assumption, no deadlock.
//   * K-loop schedule = v6's: the two wave halves run one barrier interval apart (MEM of one half beside MMA of the other),
//     4-stage filter ring, counted vmcnt, patch double-buffered per channel block.  AHEAD = how many K-steps the filter requests run
//     ahead of the MFMAs: 2 in round 2; 3 since round 3 (filter tile s + 3 lands in the stage of tile s - 1 -- see conv_igemm_v6_kernel
//     for why every wave then completes its fragment reads before the barrier that ends its MEM phase).
//   * a finisher whose producer never publishes (bounded spin: a hung or preempted block must not hang the GPU) sets ctl->error and
//     POISONS its tile with NaN; the flag is sticky, so every later launch on that workspace writes NaN too until the owner calls
//     y3_conv_workspace_reset -- a lost hand-off is loud, never a silently wrong sum (round-2 advisor finding).
//
// LDS: [4 x 16 KiB filter ring][2 x XP x 8 KiB patch][64 B zeros]; the epilogue re-uses [0, 128 KiB) as per-wave transpose slices.

#include <utility>

constexpr int V7_W_STAGE = 256 * 32 * 2;       // one filter stage: 256 filters x 32 k x 2 B
constexpr int V7_RING = 4 * V7_W_STAGE;
constexpr int V7_MAX_BLOCKS = 256;
constexpr size_t V7_HDR_BYTES = 64 + 4 * 1024;                 // control words + arrival flags
constexpr size_t V7_SLAB_BYTES = (size_t)256 * 256 * 4;        // one fp32 accumulator tile

struct V7Ctl {
    unsigned ticket, finished, error, pad[13];
};

template <int I> struct IC {
    static constexpr int value = I;
};
template <typename F, int... Is> Y3_DEV void static_for_impl(F&& f, std::integer_sequence<int, Is...>) { (f(IC<Is>{}), ...); }
template <int N, typename F> Y3_DEV void static_for(F&& f) { static_for_impl(f, std::make_integer_sequence<int, N>{}); }

Y3_DEV void wait_vm(int n) {   // n is wave-uniform; the immediate must be a literal
    switch (n) {
        case 0: wait_vmcnt<0>(); break;
        case 1: wait_vmcnt<1>(); break;
        case 2: wait_vmcnt<2>(); break;
        case 3: wait_vmcnt<3>(); break;
        case 4: wait_vmcnt<4>(); break;
        case 5: wait_vmcnt<5>(); break;
        default: wait_vmcnt<6>(); break;
    }
}

#ifdef Y3_TIMELINE   // debug build (tools/v7_probe.py): per-wave cycle sums of the four intervals of a K-step
#define V7_T(i) do { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); tsum[i] += t_ - tprev; tprev = t_; } while (0)
#else
#define V7_T(i) do { } while (0)
#endif

// ABL (tools/v7_ablate.py, -DY3_ABLATE builds only; 0 in the shipped library): what a K-step costs without one of its parts -- results are
// garbage, only the launch time means something.  1: half the MFMAs; 2: no filter requests in the loop; 3: no patch requests; 4: no pixel
// fragment reads; 5: no fragment reads at all; 6: no MFMAs; 7: no requests and no reads (MMA + barriers only)
template <typename T, int XP, int AHEAD, int ABL = 0>
__global__ __launch_bounds__(512, 2) void conv_igemm_v7_kernel(const ConvArgs p) {
#if defined(__HIP_DEVICE_COMPILE__)
#ifdef Y3_TIMELINE
    unsigned long long tsum[6] = {0, 0, 0, 0, 0, 0}, tprev = 0;
#endif
    constexpr int MC = 2, MP = 4;
    constexpr int PATCH_BYTES = XP * 8 * 1024;
    constexpr int PATCH_OFF = V7_RING;
    constexpr int ZOFF = (V7_RING + 2 * PATCH_BYTES) > 131072 ? (V7_RING + 2 * PATCH_BYTES) : 131072;
    constexpr int LDS_BYTES = ZOFF + 128;   // [ZOFF, +64): the zero block of the edge taps; [ZOFF + 64, +128): block-wide broadcast words
    typedef typename Mfma<T>::frag frag;
    typedef __attribute__((address_space(3))) void* lds_ptr_t;

    __shared__ __attribute__((aligned(64))) unsigned char smem[LDS_BYTES];   // the ONLY LDS object (guide: a second one de-pipelines the DMA waits)

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wc = wv >> 1, wp = wv & 1;   // 4 filter waves x 2 pixel waves, wave tile 64 filters x 128 pixels
    const int half = wv >> 2;              // 0: leading half, 1: trailing half (one barrier interval behind)
    const int G = gridDim.x;

    V7Ctl* ctl = (V7Ctl*)p.ws;
    unsigned* flags = (unsigned*)((char*)p.ws + 64);
    float* slabs = (float*)((char*)p.ws + V7_HDR_BYTES);

    // Work assignment.  Blocks take a ticket in START order.  Ticket t belongs to group t % NG (the XCD the dispatcher is
    // observed to place it on: speed only) and is block t / NG of that group.  A group owns a rectangle of whole tiles (see
    // below); inside a group the (tile, channel block) units are split evenly over its blocks.  Dependencies never cross groups
    // and only point at EARLIER tickets (see the item loop), so progress needs no residency assumption.
    unsigned* bcast = (unsigned*)(smem + ZOFF + 64);
    if (tid == 0) {
        bcast[0] = atomicAdd(&ctl->ticket, 1u);
        bcast[1] = __hip_atomic_load(&ctl->error, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // sticky: a lost hand-off of an EARLIER launch on this workspace
    }
    __syncthreads();
    const int ticket = __builtin_amdgcn_readfirstlane(bcast[0]);
    bool poisoned = __builtin_amdgcn_readfirstlane(bcast[1]) != 0u;   // this block writes its tiles as NaN
    __syncthreads();
    if (tid < 4) *(u32x4*)(smem + ZOFF + tid * 16) = u32x4{0u, 0u, 0u, 0u};
    __syncthreads();
    const int NG = G < 8 ? G : 8;
    const int xg = ticket % NG, ig = ticket / NG;
    const int nblk = G / NG + (xg < G % NG ? 1 : 0);
    const int ncb = p.cin_blocks;
    // A group's tiles form a rectangle (range of pixel tiles) x (range of filter tiles): gc filter-tile ranges x NG / gc pixel-tile ranges.
    // The host picks gc to minimise what the 8 private L2s fetch together: filters x (NG / gc) + activations x gc (launch_v7).
    const int gc = (NG == 8) ? p.v7_gc : 1, gp = NG / gc;
    const int ctg = xg / gp, ptg = xg - ctg * gp;
    const int nct_g = p.n_ct / gc;                       // filter tiles of this group (gc divides n_ct)
    const int ct0 = ctg * nct_g;
    const int pt0 = (int)((long long)p.n_pt * ptg / gp), pt1 = (int)((long long)p.n_pt * (ptg + 1) / gp);
    const int gtiles = (pt1 - pt0) * nct_g;               // tiles of this group, filter tile fastest
    const int Ug = gtiles * ncb;                          // units of this group
    int u_lo = (int)((long long)Ug * ig / nblk);          // this block's range, group-local
    int hi = (int)((long long)Ug * (ig + 1) / nblk);
    if (p.v7_whole) {                                     // whole tiles only: no tile is shared between blocks, no slabs
        u_lo = (int)((long long)gtiles * ig / nblk) * ncb;
        hi = (int)((long long)gtiles * (ig + 1) / nblk) * ncb;
    }

    const auto rsrc_x = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, (int)p.x_bytes, 0x00020000);
    const auto rsrc_w = __builtin_amdgcn_make_buffer_rsrc((void*)p.w, 0, (int)p.w_bytes, 0x00020000);

    // Items are visited from the END of the range to its start: a tile whose tail belongs to the next block is computed FIRST and
    // published as an fp32 partial slab; a tile whose head belongs to previous blocks is computed LAST and completed with their
    // slabs, which were published when those (earlier-ticket) blocks started.
    while (hi > u_lo) {
        const int tl = (hi - 1) / ncb;                    // group-local tile
        const int lo = max(u_lo, tl * ncb);
        const int cb0 = lo - tl * ncb;
        const int ncbs = hi - lo;
        const bool final_part = hi == (tl + 1) * ncb;     // this item ends the tile's K range: it owns the epilogue (and the bias)
        const int ptl = tl / nct_g;
        const int pt = pt0 + ptl, ct = ct0 + (tl - ptl * nct_g);
        const int m0 = pt * 256;

        // ---- per-item lane constants (also the tile-independent ones: kept kernel-wide they are what the register allocator spills) ----
        int lane_o = lane;
        asm volatile("" : "+v"(lane_o));
        const int frow = lane_o & 31, fk = lane_o >> 5;
        int aoff[MC];   // byte offset of this lane's kk = 0 filter fragment inside a ring stage (kk = 1: ^ 32)
#pragma unroll
        for (int a = 0; a < MC; ++a) {
            const int row = (wc * MC + a) * 32 + frow;
            aoff[a] = row * 64 + ((fk ^ ((row >> 2) & 3)) << 4);
        }
        const int prow0 = wp * 128 + frow;   // patch row of this lane's first pixel for tap (0, 0)
        unsigned xpre[XP];   // byte offset of this lane's 16 bytes of patch piece j at channel block 0 (0x80000000: outside the tensor -> zeros)
#pragma unroll
        for (int j = 0; j < XP; ++j) {
            const int r = (j * 8 + wv) * 16 + (lane >> 2);
            const int g = m0 - (p.W + 1) + r;
            const int lsl = (lane & 3) ^ ((r >> 2) & 3);
            xpre[j] = (g >= 0 && g < p.M) ? (unsigned)((g * p.xpitch + lsl * 8) * 2) : 0x80000000u;
        }
        unsigned woff[2];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int row = (j * 8 + wv) * 16 + (lane >> 2);
            const int lsl = (lane & 3) ^ ((row >> 2) & 3);
            woff[j] = (unsigned)(((long long)(ct * 256 + row) * p.Kpad + lsl * 8) * 2);
        }
        int ef[MP];   // edge flags of the lane's pixels: 1 top row, 2 bottom row, 4 left column, 8 right column
#pragma unroll
        for (int b = 0; b < MP; ++b) {
            const int m = m0 + wp * 128 + b * 32 + frow;
            int n, h, w;
            pix_coords(m < p.M ? m : p.M - 1, p, n, h, w);
            ef[b] = (int)(h == 0) | ((int)(h == p.H - 1) << 1) | ((int)(w == 0) << 2) | ((int)(w == p.W - 1) << 3);
        }

        f32x16 acc[MC][MP];
#pragma unroll
        for (int a = 0; a < MC; ++a)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int cb = ct * 256 + (wc * MC + a) * 32 + 8 * g + 4 * fk;
                f32x4 bz = {0.f, 0.f, 0.f, 0.f};
                if (final_part && p.bias && cb + 4 <= p.Cout) bz = *(const f32x4*)(p.bias + cb);   // the bias enters once: with the item that owns the epilogue
#pragma unroll
                for (int b = 0; b < MP; ++b)
#pragma unroll
                    for (int q = 0; q < 4; ++q) acc[a][b][4 * g + q] = bz[q];
            }

        auto dma_w = [&](int kbyte, int stage) {
#pragma unroll
            for (int j = 0; j < 2; ++j)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_w, (lds_ptr_t)(smem + stage * V7_W_STAGE + (j * 8 + wv) * 1024), 16, woff[j] + (unsigned)kbyte, 0, 0, 0);
        };
        auto dma_x = [&](int j, int cb, int buf) {
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_x, (lds_ptr_t)(smem + PATCH_OFF + buf * PATCH_BYTES + (j * 8 + wv) * 1024), 16, xpre[j] + (unsigned)(cb * 64), 0, 0, 0);
        };
        auto mma = [&](const frag (&af)[MC], const frag (&bf)[MP]) {
#pragma unroll
            for (int a = 0; a < MC; ++a)
#pragma unroll
                for (int b = 0; b < MP; ++b) acc[a][b] = Mfma<T>::run(af[a], bf[b], acc[a][b]);
        };

        // ---- prologue: patch of the first channel block, filter tiles of the first AHEAD steps ----
        auto kbyte_of = [&](int step) {   // byte offset inside a packed filter row of K-step `step` of this item: k = tap * Cin + cb * 32
            const int cbi_ = step / 9, tap_ = step - cbi_ * 9;
            return (tap_ * p.Cin + (cb0 + cbi_) * 32) * 2;
        };
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // stores of the previous item's epilogue share the counter
#pragma unroll
        for (int j = 0; j < XP; ++j) dma_x(j, cb0, 0);
        dma_w(kbyte_of(0), 0);
        dma_w(kbyte_of(1), 1);   // nsteps >= 9
        if constexpr (AHEAD == 3) {
            dma_w(kbyte_of(2), 2);
            wait_vmcnt<4>();
        } else {
            wait_vmcnt<2>();
        }
        __builtin_amdgcn_s_barrier();              // patch + filter tile 0 visible to everyone (and the zero block)
        if (half) __builtin_amdgcn_s_barrier();    // stagger
        int s = 0;
#ifdef Y3_TIMELINE
        tprev = __builtin_amdgcn_s_memtime();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#endif
        for (int cbi = 0; cbi < ncbs; ++cbi) {
            const int cb = cb0 + cbi;
            const int buf = cbi & 1;
            const bool more_cb = cbi + 1 < ncbs;
            static_for<9>([&](auto TAP) {
                constexpr int tap = decltype(TAP)::value;
                constexpr int dh = tap / 3, dw = tap % 3;
                constexpr int MASK = (dh == 0 ? 1 : dh == 2 ? 2 : 0) | (dw == 0 ? 4 : dw == 2 ? 8 : 0);
                // ---- MEM(s): request a patch piece / filter tile s + AHEAD, read the fragments of step s, retire this wave's pieces of step s + 1 ----
                int issued = 0;   // requests younger than filter tile s + 1: they may stay in flight
                if constexpr (AHEAD == 3) {   // what MEM(s - 1) requested: filter tile s + 2 and, in taps 2 .. XP + 1 (same channel block), a patch piece
                    if ((more_cb || tap + 2 < 9) && ABL != 2 && ABL != 7) issued += 2;
                    if constexpr (tap >= 2 && tap <= XP + 1 && ABL != 3 && ABL != 7) {
                        if (more_cb) issued += 1;
                    }
                }
                if constexpr (tap >= 1 && tap <= XP && ABL != 3 && ABL != 7) {
                    if (more_cb) { dma_x(tap - 1, cb + 1, buf ^ 1); issued += 1; }
                }
                if ((more_cb || tap + AHEAD < 9) && ABL != 2 && ABL != 7) {
                    constexpr int tapa = (tap + AHEAD) % 9;
                    const int cba = cb + (tap + AHEAD >= 9 ? 1 : 0);
                    dma_w((tapa * p.Cin + cba * 32) * 2, (s + AHEAD) & 3);
                    issued += 2;
                }
                frag a0[MC], a1[MC], b0[MP], b1[MP];
                if constexpr (ABL == 4 || ABL == 5 || ABL == 7) {   // fragments from nowhere (uninitialised registers kept opaque)
#pragma unroll
                    for (int a = 0; a < MC; ++a) asm volatile("" : "=v"(a0[a]), "=v"(a1[a]));
#pragma unroll
                    for (int b = 0; b < MP; ++b) asm volatile("" : "=v"(b0[b]), "=v"(b1[b]));
                }
                if constexpr (ABL != 5 && ABL != 7) {
                    const unsigned char* wl = smem + (s & 3) * V7_W_STAGE;
#pragma unroll
                    for (int a = 0; a < MC; ++a) {
                        a0[a] = *(const frag*)(wl + aoff[a]);
                        a1[a] = *(const frag*)(wl + (aoff[a] ^ 32));
                    }
                    // (the asm statements keep these per-step: hoisted out of the channel-block loop, the 9 x 4 addresses and lane masks of
                    //  all taps would cost ~40 VGPRs + ~64 SGPRs and spill -- scratch traffic would also break the counted vmcnt)
                    int rowv = prow0;
                    asm volatile("" : "+v"(rowv));
                    const int row = rowv + dh * p.W + dw;
                    const int pa = PATCH_OFF + buf * PATCH_BYTES + row * 64 + ((fk ^ ((row >> 2) & 3)) << 4);
#pragma unroll
                    for (int b = 0; b < (ABL == 4 ? 0 : MP); ++b) {
                        int ab = pa + b * 2048;
                        if constexpr (MASK != 0) {
                            int e = ef[b];
                            asm volatile("" : "+v"(e));
                            if ((e & MASK) != 0) ab = ZOFF;
                        }
                        b0[b] = *(const frag*)(smem + ab);
                        b1[b] = *(const frag*)(smem + (ab ^ 32));
                    }
                }
                V7_T(0);   // MEM issue (requests + fragment reads issued, reads landed)
                wait_vm(issued);
                V7_T(1);   // vmcnt wait
                if constexpr (AHEAD == 3) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // stage (s - 1) & 3 is requested into by the other half right behind this barrier
                __builtin_amdgcn_sched_barrier(0);
                __builtin_amdgcn_s_barrier();
                V7_T(2);   // barrier after MEM
                // ---- MMA(s) ----
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (ABL != 8 && ABL != 9 && ABL != 10) __builtin_amdgcn_s_setprio(1);
                if constexpr (ABL == 9) __builtin_amdgcn_s_setprio(0);
                if constexpr (ABL == 10) __builtin_amdgcn_s_setprio(1);
                if constexpr (ABL != 6) mma(a0, b0);
                if constexpr (ABL != 6 && ABL != 1) mma(a1, b1);
                if constexpr (ABL == 1 || ABL == 6) {   // the unused fragments stay live up to here
#pragma unroll
                    for (int a = 0; a < MC; ++a) asm volatile("" :: "v"(a0[a]), "v"(a1[a]));
#pragma unroll
                    for (int b = 0; b < MP; ++b) asm volatile("" :: "v"(b0[b]), "v"(b1[b]));
                }
                if constexpr (ABL != 8 && ABL != 9 && ABL != 10) __builtin_amdgcn_s_setprio(0);
                if constexpr (ABL == 9) __builtin_amdgcn_s_setprio(1);   // priority to the MEM phase that follows
                if constexpr (ABL == 10) __builtin_amdgcn_s_setprio(2);
                __builtin_amdgcn_sched_barrier(0);
                V7_T(3);   // MMA issue (the last MFMAs may still be in the pipe)
                __builtin_amdgcn_s_barrier();
                V7_T(4);   // barrier after MMA
                ++s;
            });
        }
        if (!half) __builtin_amdgcn_s_barrier();   // re-align the halves: every fragment read has retired, no DMA is in flight

        auto finish = [&](f32x16 (&r)[MC][MP]) {
            // two passes of 64 pixels per wave: half the residual / offset registers of one 128-pixel pass, and the 8 x 8 KiB
            // transpose slices stay inside the filter ring
#pragma unroll
            for (int hb = 0; hb < 2; ++hb) {
                f32x16 part[MC][2];
#pragma unroll
                for (int a = 0; a < MC; ++a) { part[a][0] = r[a][2 * hb]; part[a][1] = r[a][2 * hb + 1]; }
                epilogue_wave<T, MC, 2>(p, part, smem + wv * (2 * 32 * MC * 64), ct * 256 + wc * MC * 32, m0 + wp * 128 + hb * 64, lane, (pt * 2 + wp) * 2 + hb);
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();   // the slice is private to the wave: its reads of pass 0 precede the writes of pass 1
            }
            __syncthreads();   // the slices are free again before the next item's DMA lands
        };

        // a hand-off was lost (now or in an earlier launch on this workspace): the tile goes out as NaN -- loud, not silently wrong.
        // A plain store loop on the cold path: nothing of it is live across the K loop.
        auto poison_tile = [&]() {
            const int m = m0 + (tid >> 1);
            if (m < p.M) {
                int n, ho, wo;
                pix_coords(m, p, n, ho, wo);
                const long long o = out_pix(n, ho, wo, p);
                const unsigned nan2 = sizeof(T) == 2 && std::is_same<T, f16_t>::value ? 0x7e007e00u : 0x7fc07fc0u;
                const auto rsrc_y = __builtin_amdgcn_make_buffer_rsrc((void*)p.y, 0, (int)p.y_bytes, 0x00020000);
                for (int c = ct * 256 + (tid & 1) * 128, e = c + 128; c < e; c += 8)
                    if (c + 8 <= p.Cout) __builtin_amdgcn_raw_buffer_store_b128(u32x4{nan2, nan2, nan2, nan2}, rsrc_y, (unsigned)((o * p.ypitch + c) * 2), 0, 0);
            }
            __syncthreads();
        };

        if (cb0 == 0 && final_part) {
            if (poisoned) poison_tile(); else finish(acc);   // whole tile in one item: straight from the registers
        } else {
            // ---- a share of a tile: the fp32 partial goes to this block's slab (buffer stores with scalar offsets: 32 flat pointers per
            //      lane would cost 64 VGPRs) ----
            // a block can be the producer of one tile (first item) AND the finisher of another (last item): the finisher's own share
            // goes to a second, private slab so that the published one is never overwritten
            const auto rsrc_own = __builtin_amdgcn_make_buffer_rsrc((void*)(slabs + (size_t)(final_part ? V7_MAX_BLOCKS + ticket : ticket) * 65536), 0, (int)V7_SLAB_BYTES, 0x00020000);
#pragma unroll
            for (int a = 0; a < MC; ++a)
#pragma unroll
                for (int b = 0; b < MP; ++b)
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const f32x4 v = {acc[a][b][4 * g], acc[a][b][4 * g + 1], acc[a][b][4 * g + 2], acc[a][b][4 * g + 3]};
                        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rsrc_own, tid * 16, ((a * MP + b) * 4 + g) * 8192, 0);
                    }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (!final_part) {
                // the tile continues in the next block: publish (guide G16: stores drained by every wave, barrier, one-lane release, flag)
                if (tid == 0) {
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    __hip_atomic_store(&flags[ticket], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            } else {
                // the tile ends here: its head belongs to the blocks before this one (tickets - NG, - 2 NG, ...), published when they
                // started.  Sum own + their slabs in a fixed order, then the normal epilogue.  (Adding the slabs into the live K-loop
                // accumulators instead made the compiler keep two copies of the tile and spill ~300 registers.)
                const int t_start = tl * ncb;
                int j_last = ig;   // first (lowest) contributing block
                for (int j = ig - 1; j >= 0; --j) {
                    const int uj1 = (int)((long long)Ug * (j + 1) / nblk);
                    if (uj1 <= t_start) break;
                    j_last = j;
                }
                if (tid == 0) {
                    unsigned lost = 0u;
                    for (int j = ig - 1; j >= j_last && !lost; --j) {
                        const int uj0 = (int)((long long)Ug * j / nblk), uj1 = (int)((long long)Ug * (j + 1) / nblk);
                        if (uj0 == uj1) continue;
                        unsigned spins = 0;
                        while (__hip_atomic_load(&flags[j * NG + xg], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u) {
                            __builtin_amdgcn_s_sleep(8);
                            if (++spins > (1u << 22)) { lost = 1u; break; }   // bounded: a lost producer must not hang the GPU
                        }
                    }
                    if (lost) __hip_atomic_store(&ctl->error, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // sticky until y3_conv_workspace_reset
                    bcast[2] = lost;
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                }
                __syncthreads();
                if (__builtin_amdgcn_readfirstlane(bcast[2]) != 0u) poisoned = true;   // the sum would be garbage: this tile and the block's remaining ones go out as NaN
                f32x16 r[MC][MP];
#pragma unroll
                for (int a = 0; a < MC; ++a)
#pragma unroll
                    for (int b = 0; b < MP; ++b)
#pragma unroll
                        for (int g = 0; g < 4; ++g) {
                            const f32x4 v = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc_own, tid * 16, ((a * MP + b) * 4 + g) * 8192, 0));
#pragma unroll
                            for (int q = 0; q < 4; ++q) r[a][b][4 * g + q] = v[q];
                        }
                for (int j = ig - 1; j >= j_last; --j) {
                    const int uj0 = (int)((long long)Ug * j / nblk), uj1 = (int)((long long)Ug * (j + 1) / nblk);
                    if (uj0 == uj1) continue;
                    const auto rsrc_s = __builtin_amdgcn_make_buffer_rsrc((void*)(slabs + (size_t)(j * NG + xg) * 65536), 0, (int)V7_SLAB_BYTES, 0x00020000);
#pragma unroll
                    for (int a = 0; a < MC; ++a)
#pragma unroll
                        for (int b = 0; b < MP; ++b)
#pragma unroll
                            for (int g = 0; g < 4; ++g) {
                                const f32x4 v = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc_s, tid * 16, ((a * MP + b) * 4 + g) * 8192, 0));
#pragma unroll
                                for (int q = 0; q < 4; ++q) r[a][b][4 * g + q] += v[q];
                            }
                }
                __syncthreads();
                if (tid == 0)
                    for (int j = ig - 1; j >= j_last; --j) __hip_atomic_store(&flags[j * NG + xg], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // re-armed for the next launch
                if (poisoned) poison_tile(); else finish(r);
            }
        }
        hi = lo;
    }

#ifdef Y3_TIMELINE
    if (p.tl && lane == 0 && blockIdx.x < 64) {
#pragma unroll
        for (int i = 0; i < 6; ++i) p.tl[((long long)blockIdx.x * 8 + wv) * 8 + i] = tsum[i];
    }
#endif
    // last block out re-arms the ticket for the next launch on this workspace
    __syncthreads();
    if (tid == 0) {
        const unsigned done = atomicAdd(&ctl->finished, 1u);
        if (done == (unsigned)G - 1u) {
            __hip_atomic_store(&ctl->finished, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&ctl->ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
#endif
}

static int v7_cu_count() { return y3_cu_count(); }

// patch pieces per wave for an image width (rows needed: 256 + 2 W + 2, one piece = 16 rows, 8 waves): 0 = too wide for v7
static int v7_xp(int W) {
    const int rows = 258 + 2 * W;
    const int xp = (rows + 127) / 128;
    return xp < 3 ? 3 : (xp <= 5 ? xp : 0);
}

static bool v7_eligible(const ConvArgs& a) {
    const int mode = (int)y3_knob(Y3K_CONV_V7);   // 1: where it measured ahead; 0: never; 2: every shape the kernel can run
    if (mode == 0 || a.ups || !a.ws || a.ws_bytes < V7_HDR_BYTES + 2 * V7_MAX_BLOCKS * V7_SLAB_BYTES) return false;
    if (a.ks != 3 || a.stride != 1 || a.pad != 1 || a.dil_shift != 0 || a.ntaps != 9 || a.omul != 1 || a.ooh != 0 || a.oow != 0) return false;
    if (a.H != a.Ho || a.W != a.Wo || a.oH != a.Ho || a.oW != a.Wo) return false;
    if ((a.Cin % 32) != 0 || (a.Cout % 256) != 0 || v7_xp(a.W) == 0) return false;
    if (a.Cin < 256 && mode != 2) return false;   // K = 1152 (4 channel blocks per tile): the per-tile prologue / epilogue outweighs the faster main loop (169 vs 153 us @80x80)
    if (!a.x_bytes || !a.w_bytes || !a.y_bytes || (a.res && !a.r_bytes)) return false;
    for (int t = 0; t < 9; ++t)
        if (a.tdh[t] != t / 3 || a.tdw[t] != t % 3) return false;
    const long long units = (long long)y3_ceil_div(a.M, 256) * (a.Cout / 256) * (a.Cin / 32);
    return units >= 32 && units < 0x7fffffffLL;   // tiny problems stay on the one-tile-per-block kernels
}

template <typename T> int launch_v7(ConvArgs& a, hipStream_t st) {
    a.n_ct = a.Cout / 256;
    a.n_pt = y3_ceil_div(a.M, 256);
    set_divisors(a);
    a.cin_blocks = a.Cin / 32;
    a.nk = 9 * a.cin_blocks;
    a.stat_wp = 4;   // statistics rows per pixel tile: 2 pixel waves x 2 epilogue passes
    g_last_variant = "v7";
    if (a.dry) return 0;
    const long long units = (long long)a.n_ct * a.n_pt * a.cin_blocks;
    long long g = v7_cu_count();
    if (g > V7_MAX_BLOCKS) g = V7_MAX_BLOCKS;
    if (g > units / 4) g = units / 4 > 0 ? units / 4 : 1;   // at least 4 channel blocks (36 K-steps) per block: below that the slab traffic outweighs the parallelism
    // Whole tiles per block unless the launch has too few tiles to occupy the chip.  Measured on MI355X (profiles/r02_conv_v7.md): at
    // 200-800 tiles the fp32 slab round trips of an even K split (256 KiB per cut tile through a CU's ~25 GB/s store path) cost more
    // than the 22 % of idle CU-rounds they recover (172 vs 120 us on 512->1024 @20x20, batch 32); with a handful of tiles (small
    // batches) the split is what puts every CU to work.
    const long long tiles = (long long)a.n_ct * a.n_pt;
    a.v7_whole = tiles >= 64 ? 1 : 0;
    {   // filter-tile ranges per XCD group: minimise filters x (8 / gc) + activations x gc over the divisors of n_ct
        const double wbytes = (double)a.Cout * a.Kpad * 2.0, xbytes = (double)a.M * a.Cin * 2.0;
        int best = 1;
        double best_cost = wbytes * 8 + xbytes;
        for (int gc = 2; gc <= 8; gc *= 2) {
            if (a.n_ct % gc) break;
            const double cost = wbytes * (8 / gc) + xbytes * gc;
            if (cost < best_cost) { best = gc; best_cost = cost; }
        }
        a.v7_gc = best;
        const int v = (int)y3_knob(Y3K_V7_GC);
        if (v >= 1 && v <= 8 && (8 % v) == 0 && (a.n_ct % v) == 0) a.v7_gc = v;
    }
    {   // knob "v7_grid": N > 0 caps the grid; -1 = whole tiles; -2 = even K split (stream-K) whatever the tile count
        const long long g_env = y3_knob(Y3K_V7_GRID);
        if (g_env > 0 && g_env < g) g = g_env;
        if (g_env == -1) a.v7_whole = 1;
        if (g_env == -2) a.v7_whole = 0;
    }
    // (whole-tile mode keeps the full grid even when tiles < CUs: the 8 groups own 24..26 tiles each, a grid cut to the tile count would
    //  leave a 26-tile group with 25 blocks and double the makespan -- measured 193 vs 115 us)
    const int xp = v7_xp(a.W);
    const dim3 grid((unsigned)g), block(512);
#ifdef Y3_ABLATE
    if (const char* e = getenv("Y3_V7_ABL")) {   // lab build only: the environment is read per launch on purpose
        const int abl = atoi(e);
        if (xp == 3 && abl >= 1 && abl <= 10) {
            switch (abl) {
                case 1: hipLaunchKernelGGL((conv_igemm_v7_kernel<T, 3, 2, 1>), grid, block, 0, st, a); break;
                case 2: hipLaunchKernelGGL((conv_igemm_v7_kernel<T, 3, 2, 2>), grid, block, 0, st, a); break;
                case 3: hipLaunchKernelGGL((conv_igemm_v7_kernel<T, 3, 2, 3>), grid, block, 0, st, a); break;
                case 4: hipLaunchKernelGGL((conv_igemm_v7_kernel<T, 3, 2, 4>), grid, block, 0, st, a); break;
                case 5: hipLaunchKernelGGL((conv_igemm_v7_kernel<T, 3, 2, 5>), grid, block, 0, st, a); break;
                case 6: hipLaunchKernelGGL((conv_igemm_v7_kernel<T, 3, 2, 6>), grid, block, 0, st, a); break;
                case 8: hipLaunchKernelGGL((conv_igemm_v7_kernel<T, 3, 2, 8>), grid, block, 0, st, a); break;
                case 9: hipLaunchKernelGGL((conv_igemm_v7_kernel<T, 3, 2, 9>), grid, block, 0, st, a); break;
                case 10: hipLaunchKernelGGL((conv_igemm_v7_kernel<T, 3, 2, 10>), grid, block, 0, st, a); break;
                default: hipLaunchKernelGGL((conv_igemm_v7_kernel<T, 3, 2, 7>), grid, block, 0, st, a); break;
            }
            Y3_CHECK_LAUNCH();
            return 0;
        }
    }
#endif
    if (y3_knob(Y3K_CONV_AHEAD) == 2) {
        if (xp == 3) hipLaunchKernelGGL((conv_igemm_v7_kernel<T, 3, 2>), grid, block, 0, st, a);
        else if (xp == 4) hipLaunchKernelGGL((conv_igemm_v7_kernel<T, 4, 2>), grid, block, 0, st, a);
        else hipLaunchKernelGGL((conv_igemm_v7_kernel<T, 5, 2>), grid, block, 0, st, a);
    } else {
        if (xp == 3) hipLaunchKernelGGL((conv_igemm_v7_kernel<T, 3, 3>), grid, block, 0, st, a);
        else if (xp == 4) hipLaunchKernelGGL((conv_igemm_v7_kernel<T, 4, 3>), grid, block, 0, st, a);
        else hipLaunchKernelGGL((conv_igemm_v7_kernel<T, 5, 3>), grid, block, 0, st, a);
    }
    Y3_CHECK_LAUNCH();
    return 0;
}
