// conv_v10.h -- included by conv.hip INSIDE its anonymous namespace (shares ConvArgs, Mfma, epilogue_wave, fdiv, IC / static_for, ...).
//
// v10: the 3x3 / stride 1 / pad 1 convolutions with Cout % 256 == 0 and Cin % 32 == 0 (reference models/common.py:57-81 Conv inside
// Bottleneck.cv2, models/yolov3.yaml:23-31 and the 3x3 convs of the head; their data gradients run through the same kernel on the flipped
// bank) with ONE wave per SIMD, PERSISTENT blocks and the filter operand in registers.  It keeps what round 3's v9 measured as right --
// 4 waves, wave tile 64 filters x MP*32 pixels with the accumulators in AGPRs, the pixel operand as a halo patch in PADDED-IMAGE order
// with an 80-byte row pitch (position Q(n, h, w) = (n (H + 1) + h + 1) (W + 2) + w + 1: one zero row between images, one zero column on
// either side of a row, the zeros landed by out-of-range `buffer_load ... lds` lanes; tap (dh, dw) of EVERY pixel is patch row
// r(m) + dh (W + 2) + dw, so dw * 80 and the k-substep are instruction immediates and there are no edge masks) -- and changes the three
// things its ablation (profiles/r03_v9_ablation.txt) priced:
//   * FILTER FRAGMENTS BY REGISTER LOADS.  The packed bank of an eligible layer carries a second, fragment-ordered copy
//     (y3_frag_index, y3_common.h): the 4 KiB a wave needs for one K-step are contiguous and lane-ordered, so a K-step's filter operand is
//     four 1 KiB `buffer_load_dwordx4` into a ring of three register sets, two K-steps ahead.  No LDS-DMA requests, no LDS stages and no
//     `ds_read` for that operand: 14 instead of 18 fragment reads per K-step (measured -7 % as ablation arm 9).
//   * PERSISTENT BLOCKS OVER 32-PIXEL COLUMN BLOCKS.  The pixel axis is cut into column blocks of 32 pixels; the blocks of a filter tile
//     share them evenly (q or q + 1 each) and every block cuts ITS run into tiles of 6 / 7 / 8 column blocks (three bodies in one
//     kernel).  v9 planned equal tiles (200 valid of 224 computed pixels at batch 32: 12 % of the MFMA issue multiplied padding); here
//     12800 k pixels become runs of 13 (7 + 6) or 25 (7 + 6 + 6 + 6) column blocks: computed = valid.
//   * NO PER-TILE PROLOGUE.  While a block multiplies the last channel block of a tile it requests the first channel block of its NEXT
//     tile into the other patch buffer, and the filter ring simply wraps (the next tile has the same filter rows): the next tile's first
//     K-step has its operands when the epilogue ends.  The epilogue transposes through a slice of its own, so nothing waits for it.
//
//   * SMALL LAUNCHES: K SPLIT (SPLIT form, "v10k").  Below a quarter round of tiles (batch 1-4 at 640 x 640) the tiles alone do not fill the chip.  The channel
//     blocks of every tile are then cut into S slices, a block multiplies ONE slice of its tiles and writes the fp32 accumulators into slab s of the caller's
//     workspace ([slice][pixel][filter]); a second launch (conv_v10_reduce_kernel) sums the slabs in slice order and applies what the epilogue applies -- bias,
//     SiLU, rounding, statistics rows, residual.  Two launches, fixed summation order, bit-deterministic, nothing to spin on.  (It replaces round 2's stream-K
//     kernel conv_v7.h -- published slabs, arrival flags, a bounded spin and a sticky error flag for the hand-off that never came -- deleted in round 4.)
//
// LDS (144 KiB, one block per CU): [2 x 54 KiB patch buffers][4 x 1 KiB dump slots for request slots with nothing to fetch][4 x 8 KiB epilogue slices].

// HALF = false: one block per CU, bodies of 6 / 7 / 8 column blocks, 64-pixel epilogue passes.  HALF = true: TWO blocks per CU (<= 256 registers and 80 KiB of
// LDS each: the two waves of a SIMD belong to different blocks, so one block's epilogue -- VALU, LDS transpose, stores: nothing the matrix pipe does -- runs beside
// the other block's K loop), bodies of 3 / 4 column blocks, 32-pixel epilogue passes.
template <bool HALF> struct V10Geom {
    static constexpr int PB = HALF ? 31 * 1024 : 54 * 1024;       // one patch buffer
    static constexpr int DUMP = 2 * PB;                           // dump slot(s) for request slots with nothing to fetch
    static constexpr int DUMP_BYTES = HALF ? 1024 : 4 * 1024;     // (HALF: one slot shared by the four waves -- garbage over garbage)
    static constexpr int SLICE = DUMP + DUMP_BYTES;               // epilogue transpose slices, one per wave
    static constexpr int SLICE_BYTES = HALF ? 4096 : 8192;        // 32 / 64 pixels x 64 filters
    static constexpr int LDS = SLICE + 4 * SLICE_BYTES;
    static constexpr int MAXPIECE = PB / 1024;
    static constexpr int MP_LO = HALF ? 3 : 6, MP_HI = HALF ? 4 : 8;
    static constexpr int PASS = HALF ? 1 : 2;                     // column blocks per epilogue pass
};
constexpr int V10_PITCH = 80;
static_assert(V10Geom<false>::LDS <= 163840 && V10Geom<true>::LDS <= 81920, "the LDS of a CU / half of it");

// `s_waitcnt vmcnt(N)` through the builtin (the waitcnt pass parses it; an asm statement is invisible to it and it would add its own
// conservative waits in front of the next requests -- profiles/r02_conv_v8.md): simm16 = vmcnt[3:0] | expcnt[6:4] | lgkmcnt[11:8] | vmcnt[5:4] << 14
template <int N> Y3_DEV void v10_wait_vm() { __builtin_amdgcn_s_waitcnt((N & 15) | (7 << 4) | (15 << 8) | ((N >> 4) << 14)); }

// ABL (tools/v10_ablate.py, -DY3_ABLATE builds only; 0 in the shipped library): the kernel without one of its parts, garbage results, only the launch
// time means something.  1: no epilogue; 2: the epilogue with its stores and residual loads dropped by the descriptors' bounds check; 4: no MFMAs;
// 5: no fragment reads, no filter loads, no patch requests (MFMAs + epilogue only); 6: no filter loads; 7: no pixel-fragment reads; 8: no patch requests
#ifdef Y3_TIMELINE   // debug build (tools/v10_timeline.py): thread 0 of every block stamps the 100 MHz wall clock at block start and, per tile, at K-loop start / K-loop end / epilogue end
#define V10_STAMP(i) do { if (p.tl && threadIdx.x == 0 && (i) < 64) p.tl[(long long)blockIdx.x * 64 + (i)] = wall_clock64(); } while (0)
#else
#define V10_STAMP(i) do { } while (0)
#endif
constexpr size_t V10_WS_SLABS = 8192;   // byte offset of the fp32 slabs in the conv workspace (the first bytes were round 2's control words: left alone)

// ---- which column blocks a block computes (host + device: the kernel and y3_conv_v10_tiles -- the CPU test of the tiling -- run the same arithmetic) --------------------
// Block bi of a filter tile computes nt tiles of tq (+ 1 for the first tr) 32-pixel column blocks: q + 1 column blocks in nt_hi tiles for the first r blocks ("long
// share"), q in nt_lo tiles for the others.  WHICH column blocks: the v10_g blocks of a group (= the blocks of this filter tile on one XCD) own one contiguous range of
// the pixel axis and take its tiles round-robin -- in round t the group works on v10_g NEIGHBOURING tiles, so the two halo rows a tile shares with each neighbour are
// requested by blocks of the same XCD at the same channel-block step and meet in its L2 (a block walking a contiguous run of its own re-fetched them a tile later,
// after ~26 MB of other traffic had passed through the 4 MB).  Tile (t, block i of the group) starts behind the rounds before it and the blocks before it in its
// round; the first rg blocks of a group are the ones with the long share.  v10_g == 1 is the contiguous run (bi q + min(bi, r) + t tq + min(t, tr)).
__host__ __device__ inline int v10_imin(int a, int b) { return a < b ? a : b; }
__host__ __device__ inline int v10_imax(int a, int b) { return a > b ? a : b; }
struct V10Share {
    bool big;                     // long share
    int nt, tile_base, tq, tr;    // tiles, index of the first one among the filter tile's tiles (statistics rows), column blocks per tile (+ 1 for the first tr)
    int gs, rg, sg, i_hi, i_lo;   // group: first column block, blocks with the long / short share, blocks of either kind in front of this one
};
__host__ __device__ inline V10Share v10_share(const ConvArgs& p, int bi, int grp) {   // grp = bi / p.v10_g (the caller divides: multiply-shift on the device)
    V10Share s;
    s.big = bi < p.v10_r;
    s.nt = s.big ? p.v10_nt_hi : p.v10_nt_lo;
    s.tile_base = s.big ? bi * p.v10_nt_hi : p.v10_r * p.v10_nt_hi + (bi - p.v10_r) * p.v10_nt_lo;
    s.tq = s.big ? p.v10_tq_h : p.v10_tq_l;
    s.tr = s.big ? p.v10_tr_h : p.v10_tr_l;
    const int g0 = grp * p.v10_g, gi = bi - g0;
    const int gsz = v10_imin(p.v10_g, p.v10_B - g0);
    s.rg = v10_imax(0, v10_imin(gsz, p.v10_r - g0));
    s.sg = gsz - s.rg;
    s.gs = g0 * p.v10_q + v10_imin(g0, p.v10_r);
    s.i_hi = v10_imin(gi, s.rg);
    s.i_lo = v10_imax(gi - s.rg, 0);
    return s;
}
// tile t of the share: first column block and column blocks
__host__ __device__ inline void v10_tile_cols(const ConvArgs& p, const V10Share& s, int t, int& c0, int& sz) {
    sz = s.tq + (t < s.tr ? 1 : 0);
    const int th = v10_imin(t, p.v10_nt_hi), tl = v10_imin(t, p.v10_nt_lo);   // rounds before t in which the long / short shares had a tile
    const int hi_t = t < p.v10_nt_hi ? p.v10_tq_h + (t < p.v10_tr_h ? 1 : 0) : 0;
    const int lo_t = t < p.v10_nt_lo ? p.v10_tq_l + (t < p.v10_tr_l ? 1 : 0) : 0;
    c0 = s.gs + s.rg * (th * p.v10_tq_h + v10_imin(th, p.v10_tr_h)) + s.sg * (tl * p.v10_tq_l + v10_imin(tl, p.v10_tr_l)) + s.i_hi * hi_t + s.i_lo * lo_t;
}

template <typename T, int XQ, bool HALF, bool SPLIT = false, int ABL = 0>
__global__ __launch_bounds__(256, HALF ? 2 : 1) void conv_igemm_v10_kernel(const ConvArgs p) {
#if defined(__HIP_DEVICE_COMPILE__)
    constexpr int MC = 2;
    constexpr int NXP = 7 * XQ;   // patch request slots per wave and channel block: XQ in each of taps 0..6
    typedef typename Mfma<T>::frag frag;
    typedef __attribute__((address_space(3))) void* lds_ptr_t;
    typedef V10Geom<HALF> G;
    constexpr int V10_PB = G::PB, V10_DUMP = G::DUMP, V10_SLICE = G::SLICE;
    __shared__ __attribute__((aligned(1024))) unsigned char smem[G::LDS];   // the ONLY LDS object

    const int tid = threadIdx.x;
    const int lane0 = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int PW = p.W + 2;
    const int ncb = p.cin_blocks;

    // ---- this block's share: filter tile ct (slowest: the blocks of one XCD share a filter tile, its rows stay in that L2) and a run of column blocks
    int lin = xcd_remap(blockIdx.x, gridDim.x);
    int cb0 = 0, cb1 = ncb, slice = 0;   // SPLIT: this block's slice of the channel blocks (the slices of one (filter tile, run) are `per` block ids apart)
    if constexpr (SPLIT) {
        const int per = p.n_ct * p.v10_B;
        slice = fdiv(lin, p.dv_sl_mul, p.dv_sl_sh);
        lin -= slice * per;
        const int cq = ncb / p.v10_S, cr = ncb - cq * p.v10_S;
        cb0 = slice * cq + (slice < cr ? slice : cr);
        cb1 = cb0 + cq + (slice < cr ? 1 : 0);
    }
    const int ct = fdiv(lin, p.dv_ct_mul, p.dv_ct_sh);   // host: the divisor is v10_B here
    const int bi = lin - ct * p.v10_B;
    const V10Share sh = v10_share(p, bi, fdiv(bi, p.dv_g_mul, p.dv_g_sh));   // (comment above v10_share)
    const int nt = sh.nt, tile_base = sh.tile_base, tq = sh.tq, tr = sh.tr;

    const auto rsrc_x = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, (int)p.x_bytes, 0x00020000);
    // the fragment-ordered copy follows the row-major bank
    const auto rsrc_wf = __builtin_amdgcn_make_buffer_rsrc((void*)((const unsigned char*)p.w + p.w_bytes), 0, (int)p.w_bytes, 0x00020000);
    constexpr unsigned OOB = 0x80000000u;   // stays out of range when a channel-block offset is added

    // geometry of tile t: first / end pixel, padded position of the first pixel, 1 KiB pieces of its halo patch
    auto tile_geom = [&](int t, int& m0, int& m1, int& Qf, int& npiece) {
        int c0, sz;
        v10_tile_cols(p, sh, t, c0, sz);
        m0 = c0 * 32;
        m1 = min(m0 + sz * 32, p.M);
        int n, h, w;
        pix_coords(m0, p, n, h, w);
        Qf = (n * (p.H + 1) + h + 1) * PW + w + 1;
        pix_coords(m1 - 1, p, n, h, w);
        const int Ql = (n * (p.H + 1) + h + 1) * PW + w + 1;
        npiece = ((Ql - Qf + 2 * PW + 3) * V10_PITCH + 1023) >> 10;   // patch rows Qf - PW - 1 .. Ql + PW + 1
    };

    // ---- per-lane sources of this wave's patch pieces (piece q = 4 i + wave; 64 lanes x 16 B: 12.8 patch rows of 4 data slots + 1 pad slot)
    unsigned xsrc[NXP];
    auto set_xsrc = [&](int Q0, int npiece) {
        int lane = lane0;
        asm volatile("" : "+v"(lane));
#pragma unroll
        for (int i = 0; i < NXP; ++i) {
            const int q = i * 4 + wv;
            const int e = q * 64 + lane;
            const int pos = e / 5, slot = e - pos * 5;
            const int Qa = Q0 + pos;
            const int Qc = Qa > 0 ? Qa : 0;
            const int R = (int)(__umulhi((unsigned)Qc, p.dv_pw_mul) >> (p.dv_pw_sh - 1));   // (W + 2 and H + 1 are never 1: no mul == 0 form, no branch)
            const int C = Qc - R * PW;
            const int n = (int)(__umulhi((unsigned)R, p.dv_h1_mul) >> (p.dv_h1_sh - 1));
            const int hh = R - n * (p.H + 1);
            const bool ok = (slot < 4) & (q < npiece) & (Qa >= 0) & (C >= 1) & (C <= p.W) & (hh >= 1) & (n < p.N);
            xsrc[i] = ok ? (unsigned)((((n * p.H + hh - 1) * p.W + (C - 1)) * p.xpitch + slot * 8) * 2) : OOB;
        }
    };
    // slot i of the wave -> patch buffer `buf`; go = false (wave-uniform): nothing to fetch, the piece goes to the dump slot
    auto dma_x = [&](int i, int cbyte, int buf, int npiece, bool live) {
        const int q = i * 4 + wv;
        const bool go = live && q < npiece;
        const int dst = go ? buf * V10_PB + q * 1024 : V10_DUMP + (HALF ? 0 : wv * 1024);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_x, (lds_ptr_t)(smem + dst), 16, xsrc[i], cbyte, 0, 0);
    };

    // ---- filter operand: ring of three K-steps x 4 fragments (j = 2 kk + a: k-substep kk, filter rows 32 a .. 32 a + 31 of the wave's 64), one 1 KiB load each
    u32x4 Ar[3][4];
    const int a_base = (ct * 4 + wv) * p.nk * 4096;   // this wave's stream: nk K-steps of 4 KiB in the order the loop consumes them (channel block, tap)
    const int a_wrap = p.nk * 4096;
    int a_next = 0;
    unsigned a_lane = 0;   // lane * 16, set per tile (see run_tile: nothing lane-derived is kept across a tile's epilogue)
    auto a_load = [&](auto SLOT) {
        constexpr int sl = decltype(SLOT)::value;
#pragma unroll
        for (int j = 0; j < 4; ++j) Ar[sl][j] = __builtin_amdgcn_raw_buffer_load_b128(rsrc_wf, a_lane, a_base + a_next + j * 1024, 0);
        a_next += 4096;
        if (a_next == a_wrap) a_next = 0;   // (the loads of a tile's last two K-steps fetch K-steps 0 / 1 again and are dropped: a ring kept across the epilogue is a ring spilled)
    };

    int par = 0;   // patch buffer of the current channel block (alternates per channel block, across tiles too)

    // ---- one tile of MP column blocks: K loop + epilogue.  On entry: the patch of its channel block 0 is visible in buffer `par`, xsrc are this tile's
    // sources.  On exit the same holds for the next tile (has_next).
    auto run_tile = [&](auto MPC, const int m0, const int m1, const int Qf, const int npiece, const int stat_row0, const bool has_next, const int nQf,
                        const int nnpiece) {
        constexpr int MP = decltype(MPC)::value;
        constexpr int NPASS = (MP + G::PASS - 1) / G::PASS;
        // the lane id behind an opaque move: everything derived from it (fragment rows, patch offsets, the epilogue's store pattern) is re-derived per tile
        // instead of being hoisted out of the tile loop and spilled around the K loop (first cut: 170-200 spilled registers, each reload an exposed round
        // trip in the epilogue: +40 us per tile)
        int lane = lane0;
        asm volatile("" : "+v"(lane));
        const int frow = lane & 31, fk = lane >> 5;
        // filter fragments of K-steps 0 and 1: in flight under the set-up below
        a_lane = (unsigned)lane * 16u;
        a_next = cb0 * 9 * 4096;
        a_load(IC<0>{});
        a_load(IC<1>{});
        // pixels: column block b, tap row dh -> patch row (Q(m) - Qf) + dh PW; columns beyond the tile's valid pixels re-read its last pixel
        int bb[3][MP];
#pragma unroll
        for (int b = 0; b < MP; ++b) {
            int m = m0 + b * 32 + frow;
            m = m < m1 ? m : m1 - 1;
            int n, h, w;
            pix_coords(m, p, n, h, w);
            const int r = (n * (p.H + 1) + h + 1) * PW + w + 1 - Qf;
#pragma unroll
            for (int dh = 0; dh < 3; ++dh) bb[dh][b] = par * V10_PB + (r + dh * PW) * V10_PITCH + fk * 16;
        }
        // the accumulators start at the bias of their filter (lane holds filters 8g + 4fk + q of each 32-filter tile)
        f32x16 acc[MC][MP];
#pragma unroll
        for (int a = 0; a < MC; ++a)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int cbias = ct * 256 + (wv * MC + a) * 32 + 8 * g + 4 * fk;
                f32x4 bz = {0.f, 0.f, 0.f, 0.f};   // SPLIT: the bias is added once, by the slab sum
                if constexpr (!SPLIT) bz = *(const f32x4*)(p.bias + cbias);   // (Cout % 256 == 0 and the C ABI requires a bias: no guards, no branches around the accumulators' first values)
#pragma unroll
                for (int b = 0; b < MP; ++b)
#pragma unroll
                    for (int q = 0; q < 4; ++q) acc[a][b][4 * g + q] = bz[q];
            }
        frag B0[MP], B1[MP];
#pragma unroll
        for (int b = 0; b < MP; ++b) B0[b] = *(const frag*)(smem + bb[0][b]);
        int bufd = par ? -V10_PB : V10_PB;   // what moves the pixel bases to the other patch buffer
        V10_STAMP(1 + 3 * (stat_row0 / 4 - tile_base));

        int cb = cb0;
        do {   // (at least one channel block: a zero-trip path would make the register allocator keep the accumulators' first values on the stack for it)
            const bool lastcb = cb + 1 == cb1;
            int np_req = npiece, cbyte = (cb + 1) * 64;
            bool live = true;
            if (lastcb) {   // the requests of this channel block fetch the first channel block of the block's next tile
                live = has_next;
                np_req = nnpiece;
                cbyte = cb0 * 64;
                if (has_next) set_xsrc(nQf - PW - 1, nnpiece);
            }
            const int nbuf = par ^ 1;
            static_for<9>([&](auto TAP) {
                constexpr int tap = decltype(TAP)::value;
                constexpr int dh = tap / 3, dw = tap % 3;
                constexpr int ntap = (tap + 1) % 9, ndh = ntap / 3, ndw = ntap % 3;
                // ---- phase 1: MFMAs of substep 0 | pixel fragments of substep 1, filter fragments of K-step s + 2 (ring slot (s + 2) % 3 = (tap + 2) % 3: 9 % 3 == 0)
#pragma unroll
                for (int b = 0; b < MP; ++b) {
                    if constexpr (ABL == 5 || ABL == 7) asm volatile("" : "=v"(B1[b]));
                    else B1[b] = *(const frag*)(smem + bb[dh][b] + dw * V10_PITCH + 32);
                }
                if constexpr (ABL != 5 && ABL != 6) a_load(IC<(tap + 2) % 3>{});
#pragma unroll
                for (int a = 0; a < MC; ++a)
#pragma unroll
                    for (int b = 0; b < MP; ++b) {
                        if constexpr (ABL == 4) asm volatile("" :: "v"(Ar[tap % 3][a]), "v"(B0[b]));
                        else acc[a][b] = Mfma<T>::run(__builtin_bit_cast(frag, Ar[tap % 3][a]), B0[b], acc[a][b]);
                    }
                {
#pragma unroll
                    for (int i = 0; i < MP; ++i) {
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);   // one MFMA
                        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);   // one LDS read
                    }
                    constexpr int NV1 = MC * MP - MP < 4 ? MC * MP - MP : 4;   // (the 3-column-block body has 3 MFMAs left for the 4 loads)
#pragma unroll
                    for (int i = 0; i < NV1; ++i) {
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                        __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);   // one load
                    }
                    if constexpr (4 - NV1 > 0) __builtin_amdgcn_sched_group_barrier(0x020, 4 - NV1, 0);
                    if constexpr (MC * MP - MP - 4 > 0) __builtin_amdgcn_sched_group_barrier(0x008, MC * MP - MP - 4, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
                // ---- phase 2: MFMAs of substep 1 | pixel fragments of (K-step s + 1, substep 0), patch requests of the next channel block
                // (no counted wait per K-step: the filter fragments are register loads, the compiler's own vmcnt in front of their first use -- two K-steps after
                // the load -- is exact; v9 had to retire the LDS-DMA'd filter stage of K-step s + 1 here, one K-step after its request)
                if constexpr (tap == 8) {
                    v10_wait_vm<8>();   // everything but the filter loads of taps 7 and 8: the patch pieces of the next channel block (requested in taps 0..6) have landed
                    // the next channel block: its patch pieces (requested in taps 0..6, retired by the counted waits since) become visible to the
                    // other waves, and every wave is done reading this block's buffer (its last reads, B1 above, have returned) before anyone
                    // requests into it again
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    __builtin_amdgcn_s_barrier();
#pragma unroll
                    for (int d = 0; d < 3; ++d)
#pragma unroll
                        for (int b = 0; b < MP; ++b) bb[d][b] += bufd;
                    bufd = -bufd;
                }
#pragma unroll
                for (int b = 0; b < MP; ++b) {
                    if constexpr (ABL == 5 || ABL == 7) asm volatile("" : "=v"(B0[b]));
                    else B0[b] = *(const frag*)(smem + bb[ndh][b] + ndw * V10_PITCH);
                }
                if constexpr (tap < 7 && ABL != 5 && ABL != 8) {
#pragma unroll
                    for (int x = 0; x < XQ; ++x) dma_x(tap * XQ + x, cbyte, nbuf, np_req, live);
                }
#pragma unroll
                for (int a = 0; a < MC; ++a)
#pragma unroll
                    for (int b = 0; b < MP; ++b) {
                        if constexpr (ABL == 4) asm volatile("" :: "v"(Ar[tap % 3][2 + a]), "v"(B1[b]));
                        else acc[a][b] = Mfma<T>::run(__builtin_bit_cast(frag, Ar[tap % 3][2 + a]), B1[b], acc[a][b]);
                    }
                {
#pragma unroll
                    for (int i = 0; i < MP; ++i) {
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                    }
                    if constexpr (tap < 7) {
#pragma unroll
                        for (int x = 0; x < XQ; ++x) {
                            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                            __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
                        }
                    }
                    if constexpr (MC * MP - MP - (tap < 7 ? XQ : 0) > 0) __builtin_amdgcn_sched_group_barrier(0x008, MC * MP - MP - (tap < 7 ? XQ : 0), 0);
                }
                __builtin_amdgcn_sched_barrier(0);
            });
            par ^= 1;
        } while (++cb < cb1);
        V10_STAMP(2 + 3 * (stat_row0 / 4 - tile_base));

        if constexpr (SPLIT) {   // the fp32 accumulators go into slab `slice` as they are: [pixel][filter], four consecutive filters of a pixel per 16-byte store
            float* slab = (float*)((unsigned char*)p.ws + V10_WS_SLABS) + (size_t)slice * (size_t)p.M * (size_t)p.Cout;
#pragma unroll
            for (int b = 0; b < MP; ++b) {
                const int m = m0 + b * 32 + frow;
                if (m < m1) {
                    float* row = slab + (size_t)m * p.Cout + ct * 256 + wv * 64 + 4 * fk;
#pragma unroll
                    for (int a = 0; a < MC; ++a)
#pragma unroll
                        for (int g = 0; g < 4; ++g) *(f32x4*)(row + a * 32 + 8 * g) = f32x4{acc[a][b][4 * g], acc[a][b][4 * g + 1], acc[a][b][4 * g + 2], acc[a][b][4 * g + 3]};
                }
            }
            return;
        }
        // ---- epilogue: passes of 64 pixels through the wave's own transpose slice (nothing else lives there: the patch of the next tile keeps landing)
        if constexpr (ABL == 1) {   // keep the accumulators alive, store nothing
#pragma unroll
            for (int a = 0; a < MC; ++a)
#pragma unroll
                for (int b = 0; b < MP; ++b) asm volatile("" :: "v"(acc[a][b]));
            return;
        }
        unsigned char* slice = smem + V10_SLICE + wv * G::SLICE_BYTES;
        int lane_e = lane0;   // (again opaque: the store pattern is derived here, after the K loop, not kept alive through it)
        asm volatile("" : "+v"(lane_e));
        auto passes = [&](const ConvArgs& pe) {
#pragma unroll
        for (int hb = 0; hb < NPASS; ++hb) {
            if constexpr (G::PASS == 1 || MP % 2 == 1) {
                if (G::PASS == 1 || hb == NPASS - 1) {   // a 32-pixel pass
                    f32x16 part[MC][1];
#pragma unroll
                    for (int a = 0; a < MC; ++a) part[a][0] = acc[a][hb * G::PASS];
                    epilogue_wave<T, MC, 1, false, true>(pe, part, slice, ct * 256 + wv * MC * 32, m0 + hb * G::PASS * 32, lane_e, stat_row0 + hb, m1);
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                    continue;
                }
            }
            if constexpr (G::PASS == 2) {
                f32x16 part[MC][2];
#pragma unroll
                for (int a = 0; a < MC; ++a) { part[a][0] = acc[a][2 * hb]; part[a][1] = acc[a][2 * hb + 1]; }
                epilogue_wave<T, MC, 2, false, true>(pe, part, slice, ct * 256 + wv * MC * 32, m0 + hb * 64, lane_e, stat_row0 + hb, m1);
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();   // the slice is private to the wave: its reads of one pass precede the writes of the next
            }
        }
        };
        if constexpr (ABL == 2) {   // lab: every store / residual load out of range
            ConvArgs q = p;
            q.y_bytes = 0; q.r_bytes = 0;
            passes(q);
        } else {
            passes(p);
        }
        // a tile owns four statistics rows (one per 64-pixel pass of the widest body): the passes this body does not have are zero rows
        if (p.stats != nullptr && NPASS < 4 && lane_e < 8) {
            const int c = ct * 256 + wv * MC * 32 + lane_e * 8;
            if (c + 8 <= p.Cout) {
#pragma unroll
                for (int hb = NPASS; hb < 4; ++hb) {
                    float* row = p.stats + ((long long)(stat_row0 + hb) * p.Cout + c) * 2;
#pragma unroll
                    for (int q = 0; q < 4; ++q) *(f32x4*)(row + q * 4) = f32x4{0.f, 0.f, 0.f, 0.f};
                }
            }
        }
    };

    V10_STAMP(0);
    // ---- prologue of the block: the whole patch of (tile 0, channel block 0)
    int m0, m1, Qf, npiece;
    tile_geom(0, m0, m1, Qf, npiece);
    set_xsrc(Qf - PW - 1, npiece);
#pragma unroll
    for (int i = 0; i < NXP; ++i) dma_x(i, cb0 * 64, 0, npiece, true);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();

    int t = 0;
    do {
        const bool has_next = t + 1 < nt;
        int nm0 = 0, nm1 = 1, nQf = 0, nnp = 0;
        if (has_next) tile_geom(t + 1, nm0, nm1, nQf, nnp);
        const int sz = tq + (t < tr ? 1 : 0);
        const int srow = (tile_base + t) * 4;
        if constexpr (HALF) {
            if (sz >= 4) run_tile(IC<4>{}, m0, m1, Qf, npiece, srow, has_next, nQf, nnp);
            else run_tile(IC<3>{}, m0, m1, Qf, npiece, srow, has_next, nQf, nnp);
        } else {
            if (sz >= 8) run_tile(IC<8>{}, m0, m1, Qf, npiece, srow, has_next, nQf, nnp);
            else if (sz == 7) run_tile(IC<7>{}, m0, m1, Qf, npiece, srow, has_next, nQf, nnp);
            else run_tile(IC<6>{}, m0, m1, Qf, npiece, srow, has_next, nQf, nnp);
        }
        V10_STAMP(3 + 3 * t);
        m0 = nm0; m1 = nm1; Qf = nQf; npiece = nnp;
    } while (++t < nt);
#endif
}

// The second launch of the SPLIT form: y[m][c] = round(act(bias[c] + sum over slices of slab[s][m][c])) (+ residual), slices in order; a block owns 64 pixels x 256
// filters (thread = 8 consecutive filters of one pixel per pass, 8 pixels per pass), so its statistics row -- per-filter (sum, sum of squares) of the values as
// STORED, before the residual, exactly what epilogue_wave counts -- needs no atomics: per-thread sums over the passes, then the 8 pixel lanes in lane order.
template <typename T>
__global__ __launch_bounds__(256) void conv_v10_reduce_kernel(const ConvArgs p) {
#if defined(__HIP_DEVICE_COMPILE__)
    typedef typename Mfma<T>::frag vec8;
    __shared__ float red[8][32][16];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const int c = blockIdx.y * 256 + tx * 8;
    const float* slab = (const float*)((const unsigned char*)p.ws + V10_WS_SLABS);
    const size_t sl = (size_t)p.M * (size_t)p.Cout;
    const f32x4 b0 = *(const f32x4*)(p.bias + c), b1 = *(const f32x4*)(p.bias + c + 4);
    float st0[8], st1[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) st0[q] = st1[q] = 0.0f;
    for (int pass = 0; pass < 8; ++pass) {
        const int m = blockIdx.x * 64 + pass * 8 + ty;
        if (m >= p.M) break;
        const float* src = slab + (size_t)m * p.Cout + c;
        f32x8 v = {b0[0], b0[1], b0[2], b0[3], b1[0], b1[1], b1[2], b1[3]};
        for (int s = 0; s < p.v10_S; ++s) {
            const f32x4 lo = *(const f32x4*)(src + (size_t)s * sl), hi = *(const f32x4*)(src + (size_t)s * sl + 4);
#pragma unroll
            for (int q = 0; q < 4; ++q) { v[q] += lo[q]; v[4 + q] += hi[q]; }
        }
        if (p.act == Y3_ACT_SILU) silu_vec<f32x8, 8>(v);
        u32x4 ov;
#pragma unroll
        for (int q = 0; q < 4; ++q) ov[q] = pack2<T>(v[2 * q], v[2 * q + 1]);
        vec8 o = __builtin_bit_cast(vec8, ov);
        if (p.stats != nullptr) {
#pragma unroll
            for (int q = 0; q < 8; ++q) { const float f = to_f32<T>(o[q]); st0[q] += f; st1[q] += f * f; }
        }
        if (p.res != nullptr) {   // x + cv2(cv1(x)) in fp32, rounded once
            const vec8 rr = *(const vec8*)((const T*)p.res + (size_t)m * p.rpitch + c);
#pragma unroll
            for (int q = 0; q < 4; ++q) ov[q] = pack2<T>(to_f32<T>(o[2 * q]) + to_f32<T>(rr[2 * q]), to_f32<T>(o[2 * q + 1]) + to_f32<T>(rr[2 * q + 1]));
        }
        *(u32x4*)((T*)p.y + (size_t)m * p.ypitch + c) = ov;
    }
    if (p.stats != nullptr) {
#pragma unroll
        for (int q = 0; q < 8; ++q) { red[ty][tx][q] = st0[q]; red[ty][tx][8 + q] = st1[q]; }
        __syncthreads();
        if (ty == 0) {
            float* row = p.stats + ((long long)blockIdx.x * p.Cout + c) * 2;
#pragma unroll
            for (int q = 0; q < 8; q += 2) {
                float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
#pragma unroll
                for (int y = 0; y < 8; ++y) { a0 += red[y][tx][q]; a1 += red[y][tx][8 + q]; a2 += red[y][tx][q + 1]; a3 += red[y][tx][8 + q + 1]; }
                *(f32x4*)(row + q * 2) = f32x4{a0, a1, a2, a3};
            }
        }
    }
#endif
}

static thread_local ConvArgs g_v10_dry;              // the arguments a dry run of this kernel filled (y3_conv_v10_tiles)
static thread_local bool g_v10_dry_valid = false;

// The host's plan: blocks per filter tile (B), column blocks per block (q, + 1 for the first r), the widest body whose worst-case halo patch fits the
// patch buffer, tiles per block for the two run lengths.
struct V10Plan {
    int B, q, r, mp_max, nt_hi, nt_lo, n_tiles, xq, half, S;
};
static int v10_patch_pieces(const ConvArgs& a, int mp) {   // worst case over tile positions: (vp - 1) pixels + 2 pad columns per row crossing + a zero row per image crossing + the halo
    const int vp = mp * 32, PW = a.W + 2;
    const int rc = (vp - 1 + a.W - 1) / a.W, ic = (vp - 1 + a.H * a.W - 1) / (a.H * a.W);
    const int npos = (vp - 1) + 2 * rc + PW * ic + 2 * PW + 3;
    return (npos * V10_PITCH + 1023) / 1024;
}
static bool v10_plan_form(const ConvArgs& a, V10Plan& pl, bool half, int per_cu = 0) {   // per_cu: blocks per CU (0: two for the half form, one for the full one)
    const int n_ct = a.Cout / 256, cus = y3_cu_count();
    const int CB = (a.M + 31) / 32;
    if (n_ct < 1 || CB < 1) return false;
    const int force_mp = (int)y3_knob(Y3K_V10_MP), force_b = (int)y3_knob(Y3K_V10_BLOCKS);
    const int mp_lo = half ? V10Geom<true>::MP_LO : V10Geom<false>::MP_LO, mp_hi = half ? V10Geom<true>::MP_HI : V10Geom<false>::MP_HI;
    const int maxpiece = half ? V10Geom<true>::MAXPIECE : V10Geom<false>::MAXPIECE;
    pl.half = half ? 1 : 0;
    pl.S = 1;
    pl.mp_max = 0;
    for (int mp = mp_hi; mp >= mp_lo; --mp) {
        if (force_mp >= mp_lo && force_mp <= mp_hi && mp > force_mp) continue;
        if (v10_patch_pieces(a, mp) <= maxpiece) { pl.mp_max = mp; break; }
    }
    if (!pl.mp_max) return false;
    pl.xq = v10_patch_pieces(a, pl.mp_max) > 28 ? 2 : 1;
    int B = (per_cu ? per_cu : (half ? 2 : 1)) * cus / n_ct;
    if (B < 1) B = 1;
    if (B > CB / mp_lo) B = CB / mp_lo > 0 ? CB / mp_lo : 1;   // at least one narrowest body of column blocks per block where the launch has them
    if (force_b > 0) B = force_b < CB ? force_b : CB;
    pl.B = B;
    pl.q = CB / B;
    pl.r = CB % B;
    pl.nt_lo = (pl.q + pl.mp_max - 1) / pl.mp_max;
    pl.nt_hi = (pl.q + 1 + pl.mp_max - 1) / pl.mp_max;
    pl.n_tiles = pl.r * pl.nt_hi + (B - pl.r) * pl.nt_lo;
    return true;
}
// knob v10_half: 0 one block per CU; 1 two half-size blocks per CU wherever that form fits; 2 (default) the measured choice: the half form up to 72 K-steps per
// tile (Cin <= 256), where the epilogue + tile set-up it hides are 25-40 % of a tile (profiles/r04_conv_v10_half_ab.txt: 128 -> 256 @80x80 161 -> 140 us,
// 256 -> 512 @40x40 138 -> 119 us at batch 32; 512 -> 1024 @20x20 level at batch 32 and 6 % behind at batch 64: twice the filter bytes per MFMA)
static bool v10_plan(const ConvArgs& a, V10Plan& pl) {
    const int mode = (int)y3_knob(Y3K_V10_HALF);
    const bool half = mode == 1 || (mode == 2 && a.Cin <= 256);
    if (half && v10_plan_form(a, pl, true)) return true;
    return v10_plan_form(a, pl, false);
}

// the shapes the kernel can run at all (either form)
static bool v10_shape_ok(const ConvArgs& a) {
    if (y3_knob(Y3K_CONV_V10) == 0 || a.ups) return false;
    if (a.ks != 3 || a.stride != 1 || a.pad != 1 || a.dil_shift != 0 || a.ntaps != 9 || a.omul != 1 || a.ooh != 0 || a.oow != 0) return false;
    if (a.H != a.Ho || a.W != a.Wo || a.oH != a.Ho || a.oW != a.Wo) return false;
    if (!y3_filter_has_frag(a.Cout, a.Cin, a.ks)) return false;   // Cin % 32 == 0, Cout % 256 == 0: the bank carries the fragment-ordered copy
    if (!a.x_bytes || !a.w_bytes || !a.y_bytes || (a.res && !a.r_bytes)) return false;
    if (2ll * a.w_bytes >= 0x7fffffffLL) return false;
    for (int t = 0; t < 9; ++t)
        if (a.tdh[t] != t / 3 || a.tdw[t] != t % 3) return false;
    if ((long long)(a.N + 1) * (a.H + 1) * (a.W + 2) >= 0x7fffffffLL) return false;
    if ((long long)(a.Cout / 256) * 4 * 9 * (a.Cin / 32) * 4096 >= 0x7fffffffLL) return false;
    if (a.Cin < 128 && y3_knob(Y3K_CONV_V10) != 2) return false;   // K = 288 / 576: the strip kernels and the small tiles (conv_strip.h, v3)
    return true;
}
// a quarter round of 256-pixel tiles or more: whole tiles fill the chip
static bool v10_big_enough(const ConvArgs& a) { return (long long)y3_ceil_div(a.M, 256) * (a.Cout / 256) >= 64 || y3_knob(Y3K_CONV_V10) == 2; }

static bool v10_eligible(const ConvArgs& a) {
    if (!v10_shape_ok(a) || !v10_big_enough(a)) return false;
    V10Plan pl;
    return v10_plan(a, pl);
}

// SPLIT form (small launches, needs the caller's workspace): the half-size geometry where its patch fits (more, smaller tiles), S slices of the channel blocks so that
// (filter tiles x blocks x slices) reaches the resident block count; knob v10_slices forces S (tests)
static bool v10k_plan(const ConvArgs& a, V10Plan& pl) {
    if (!v10_plan_form(a, pl, true) && !v10_plan_form(a, pl, false)) return false;
    const int ncb = a.Cin / 32, cus = y3_cu_count();
    const long long nb1 = (long long)(a.Cout / 256) * pl.B;
    const long long target = (long long)(pl.half ? 2 : 1) * cus;
    long long S = (target + nb1 - 1) / nb1;
    const long long force = y3_knob(Y3K_V10_SLICES);
    if (force > 0) S = force;
    if (S > ncb) S = ncb;
    const size_t per = (size_t)a.M * (size_t)a.Cout * 4;
    if (!a.ws || a.ws_bytes <= V10_WS_SLABS || per == 0) return false;
    const long long fit = (long long)((a.ws_bytes - V10_WS_SLABS) / per);
    if (S > fit) S = fit;
    if (S < 2 && force != 1) return false;   // one slice: nothing to split (the tile kernels take it)
    if (S < 1) return false;
    pl.S = (int)S;
    return nb1 * S <= 0x7fffffffLL;
}
static bool v10k_eligible(const ConvArgs& a) {
    if (y3_knob(Y3K_V10_KSPLIT) == 0 || !v10_shape_ok(a) || (v10_big_enough(a) && y3_knob(Y3K_V10_KSPLIT) != 2)) return false;
    V10Plan pl;
    return v10k_plan(a, pl);
}

static void v10_fill_args(ConvArgs& a, const V10Plan& pl) {
    a.n_ct = a.Cout / 256;
    a.n_pt = pl.n_tiles;
    a.v10_B = pl.B; a.v10_q = pl.q; a.v10_r = pl.r; a.v10_nt_hi = pl.nt_hi; a.v10_nt_lo = pl.nt_lo; a.v10_S = pl.S;
    a.v10_tq_h = (pl.q + 1) / pl.nt_hi; a.v10_tr_h = (pl.q + 1) % pl.nt_hi;
    a.v10_tq_l = pl.nt_lo ? pl.q / pl.nt_lo : 0; a.v10_tr_l = pl.nt_lo ? pl.q % pl.nt_lo : 0;
    // interleave group = the blocks of one filter tile that xcd_remap puts on one XCD (an eighth of the grid; the whole filter tile when it has fewer); knob v10_group: 0 = contiguous runs
    const long long nb = (long long)a.n_ct * pl.B * pl.S;
    int g = y3_knob(Y3K_V10_GROUP) ? (int)(nb / 8) : 1;
    if (g > pl.B) g = pl.B;
    if (g < 1) g = 1;
    a.v10_g = g;
    magic_u31(g, a.dv_g_mul, a.dv_g_sh);
    set_divisors(a);
    magic_u31(pl.B, a.dv_ct_mul, a.dv_ct_sh);   // this kernel divides the block id by the blocks per filter tile
    magic_u31(a.n_ct * pl.B, a.dv_sl_mul, a.dv_sl_sh);   // ... and (SPLIT) by the blocks per slice
    magic_u31(a.W + 2, a.dv_pw_mul, a.dv_pw_sh);
    magic_u31(a.H + 1, a.dv_h1_mul, a.dv_h1_sh);
    a.cin_blocks = a.Cin / 32;
    a.nk = 9 * a.cin_blocks;
}

#include "conv_v10d.h"   // the deferred-epilogue form of the same tiles (uses the geometry and helpers above)
static bool v10d_plan(const ConvArgs& a, V10Plan& pl) {
    const long long mode = y3_knob(Y3K_V10_DEFER);
    if (mode == 0 || a.stats != nullptr || a.act != Y3_ACT_SILU || a.Cin < 128) return false;
    if (mode != 2 && !(y3_knob(Y3K_V10_HALF) == 2 && a.Cin <= 256)) return false;   // the shapes the half form serves
    return v10_plan_form(a, pl, true, 1);
}

template <typename T> int launch_v10(ConvArgs& a, hipStream_t st) {
    V10Plan pl;
    if (v10d_plan(a, pl)) {
        v10_fill_args(a, pl);
        a.stat_wp = 4;
        g_last_variant = "v10d";
        if (a.dry) { g_v10_dry = a; g_v10_dry_valid = true; return 0; }
        const dim3 grid((unsigned)(a.n_ct * pl.B)), block(256);
        if (a.res) {
            if (pl.xq == 2) hipLaunchKernelGGL((conv_igemm_v10d_kernel<T, 2, true>), grid, block, 0, st, a);
            else hipLaunchKernelGGL((conv_igemm_v10d_kernel<T, 1, true>), grid, block, 0, st, a);
        } else {
            if (pl.xq == 2) hipLaunchKernelGGL((conv_igemm_v10d_kernel<T, 2, false>), grid, block, 0, st, a);
            else hipLaunchKernelGGL((conv_igemm_v10d_kernel<T, 1, false>), grid, block, 0, st, a);
        }
        Y3_CHECK_LAUNCH();
        return 0;
    }
    if (!v10_plan(a, pl)) Y3_FAIL("conv v10: no tile plan (internal)");
    v10_fill_args(a, pl);
    a.stat_wp = 4;   // statistics rows per tile: one per 64-pixel epilogue pass of the widest body (narrower bodies write zero rows)
    g_last_variant = pl.half ? "v10h" : "v10";
    if (a.dry) { g_v10_dry = a; g_v10_dry_valid = true; return 0; }
    const dim3 grid((unsigned)(a.n_ct * pl.B)), block(256);
#ifdef Y3_ABLATE
    if (const char* e = getenv("Y3_V10_ABL"); e && std::is_same<T, f16_t>::value && !pl.half) {   // lab build only (f16, one block per CU)
        typedef f16_t TT;
        const int abl = atoi(e);
#define Y3_V10_ABL_CASE(XQV, N) case N: hipLaunchKernelGGL((conv_igemm_v10_kernel<TT, XQV, false, false, N>), grid, block, 0, st, a); break;
        if (pl.xq == 2) {
            switch (abl) { Y3_V10_ABL_CASE(2, 1) Y3_V10_ABL_CASE(2, 2) Y3_V10_ABL_CASE(2, 4) Y3_V10_ABL_CASE(2, 5) Y3_V10_ABL_CASE(2, 6) Y3_V10_ABL_CASE(2, 7) Y3_V10_ABL_CASE(2, 8)
                default: hipLaunchKernelGGL((conv_igemm_v10_kernel<TT, 2, false, false, 0>), grid, block, 0, st, a); break; }
        } else {
            switch (abl) { Y3_V10_ABL_CASE(1, 1) Y3_V10_ABL_CASE(1, 2) Y3_V10_ABL_CASE(1, 4) Y3_V10_ABL_CASE(1, 5) Y3_V10_ABL_CASE(1, 6) Y3_V10_ABL_CASE(1, 7) Y3_V10_ABL_CASE(1, 8)
                default: hipLaunchKernelGGL((conv_igemm_v10_kernel<TT, 1, false, false, 0>), grid, block, 0, st, a); break; }
        }
#undef Y3_V10_ABL_CASE
        Y3_CHECK_LAUNCH();
        return 0;
    }
#endif
    if (pl.half) {
        if (pl.xq == 2) hipLaunchKernelGGL((conv_igemm_v10_kernel<T, 2, true>), grid, block, 0, st, a);
        else hipLaunchKernelGGL((conv_igemm_v10_kernel<T, 1, true>), grid, block, 0, st, a);
    } else if (pl.xq == 2) hipLaunchKernelGGL((conv_igemm_v10_kernel<T, 2, false>), grid, block, 0, st, a);
    else hipLaunchKernelGGL((conv_igemm_v10_kernel<T, 1, false>), grid, block, 0, st, a);
    Y3_CHECK_LAUNCH();
    return 0;
}

// SPLIT form: the tile kernel over (slice, filter tile, run) blocks, then the slab sum
template <typename T> int launch_v10k(ConvArgs& a, hipStream_t st) {
    V10Plan pl;
    if (!v10k_plan(a, pl)) Y3_FAIL("conv v10k: no plan (internal)");
    v10_fill_args(a, pl);
    a.n_pt = y3_ceil_div(a.M, 64);   // statistics rows: one per 64-pixel block of the slab sum
    a.stat_wp = 1;
    g_last_variant = "v10k";
    if (a.dry) { g_v10_dry = a; g_v10_dry_valid = true; return 0; }
    const dim3 grid((unsigned)(a.n_ct * pl.B * pl.S)), block(256);
    if (pl.half) {
        if (pl.xq == 2) hipLaunchKernelGGL((conv_igemm_v10_kernel<T, 2, true, true>), grid, block, 0, st, a);
        else hipLaunchKernelGGL((conv_igemm_v10_kernel<T, 1, true, true>), grid, block, 0, st, a);
    } else if (pl.xq == 2) hipLaunchKernelGGL((conv_igemm_v10_kernel<T, 2, false, true>), grid, block, 0, st, a);
    else hipLaunchKernelGGL((conv_igemm_v10_kernel<T, 1, false, true>), grid, block, 0, st, a);
    Y3_CHECK_LAUNCH();
    hipLaunchKernelGGL((conv_v10_reduce_kernel<T>), dim3((unsigned)y3_ceil_div(a.M, 64), (unsigned)a.n_ct), dim3(256), 0, st, a);
    Y3_CHECK_LAUNCH();
    return 0;
}
