// conv_v10d.h -- included by conv_v10.h in front of its launch code (shares its geometry, tile plan, helpers).
//
// "v10d": conv_v10.h's half-size tiles (64 filters x 3 / 4 column blocks of 32 pixels per wave) with ONE block per CU and the epilogue of tile t dealt out between
// the MFMAs of tile t + 1's K loop, in the same wave.  Why (profiles/r06_mfma_valu_overlap_probe.txt): on this chip whatever ANOTHER wave of a SIMD issues is added in
// full to the matrix wave's time -- the two blocks per CU of the half form hide latency, not the activation -- while independent work between a wave's OWN MFMAs is
// 55-75 % hidden.  The epilogue (two quarter-rate transcendentals per output value) is 33 / 20 % of a tile at K = 1152 / 2304.
//   * two accumulator sets: `acc` (the tile being multiplied) and `prev` (the finished tile: 2 x 4 x 16 registers each = the 256 AGPRs of a one-wave-per-SIMD block);
//   * the previous tile's activation runs in four passes of 32 pixels, one per channel block of the current tile (Cin >= 128: at least four), each pass cut into pieces
//     that sit in the 18 MFMA groups (9 taps x 2 k-substeps) of that channel block: sixteen 2-value SiLU units, every fourth followed by the pack / lane-half swap of
//     8 values, the residual (one 16-byte register load per pack, requested two units ahead in the layout the pack has: loads return in order, the counted waits
//     stay exact) and the LDS write into the pass's OWN transpose slice (4 x 4 KiB per wave);
//   * NO store inside the K loop: stores share `vmcnt` with the loads and may be acknowledged out of order -- behind a load that a counted wait is waiting for they
//     can let the wait pass early, in front of it they hold every later wait back by a store round trip.  What is left behind the K loop is the light part: read the
//     four slices back in store order, 16-byte stores (no arithmetic at all);
//   * the last tile of a block leaves through epilogue_wave as before.
// Inference form only: SiLU, no statistics rows, no K split (the host falls back to conv_v10.h otherwise).  Knob "v10_defer".

template <typename T, int XQ, bool RES>
__global__ __launch_bounds__(256, 1) void conv_igemm_v10d_kernel(const ConvArgs p) {
#if defined(__HIP_DEVICE_COMPILE__)
    constexpr int MC = 2, MPX = 4;
    constexpr int NXP = 7 * XQ;
    typedef typename Mfma<T>::frag frag;
    typedef frag vec8;
    typedef __attribute__((address_space(3))) void* lds_ptr_t;
    typedef V10Geom<true> G;
    constexpr int V10_PB = G::PB, V10_DUMP = G::DUMP, V10_SLICE = G::SLICE;
    constexpr int LDS_BYTES = V10_SLICE + 4 * 4 * 4096;   // one 4 KiB transpose slice per (wave, pass)
    static_assert(LDS_BYTES <= 163840, "the LDS of a CU");
    __shared__ __attribute__((aligned(1024))) unsigned char smem[LDS_BYTES];

    const int tid = threadIdx.x;
    const int lane0 = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int PW = p.W + 2;
    const int ncb = p.cin_blocks;

    const int lin = xcd_remap(blockIdx.x, gridDim.x);
    const int ct = fdiv(lin, p.dv_ct_mul, p.dv_ct_sh);
    const int bi = lin - ct * p.v10_B;
    const V10Share sh = v10_share(p, bi, fdiv(bi, p.dv_g_mul, p.dv_g_sh));
    const int nt = sh.nt, tq = sh.tq, tr = sh.tr;

    const auto rsrc_x = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, (int)p.x_bytes, 0x00020000);
    const auto rsrc_wf = __builtin_amdgcn_make_buffer_rsrc((void*)((const unsigned char*)p.w + p.w_bytes), 0, (int)p.w_bytes, 0x00020000);
    const auto rsrc_y = __builtin_amdgcn_make_buffer_rsrc((void*)p.y, 0, (int)p.y_bytes, 0x00020000);
    const auto rsrc_r = __builtin_amdgcn_make_buffer_rsrc((void*)(RES ? p.res : p.y), 0, RES ? (int)p.r_bytes : 0, 0x00020000);
    constexpr unsigned OOB = 0x80000000u;
    constexpr unsigned OOBS = 0xffffffffu;

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
        npiece = ((Ql - Qf + 2 * PW + 3) * V10_PITCH + 1023) >> 10;
    };

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
            const int R = (int)(__umulhi((unsigned)Qc, p.dv_pw_mul) >> (p.dv_pw_sh - 1));
            const int C = Qc - R * PW;
            const int n = (int)(__umulhi((unsigned)R, p.dv_h1_mul) >> (p.dv_h1_sh - 1));
            const int hh = R - n * (p.H + 1);
            const bool ok = (slot < 4) & (q < npiece) & (Qa >= 0) & (C >= 1) & (C <= p.W) & (hh >= 1) & (n < p.N);
            xsrc[i] = ok ? (unsigned)((((n * p.H + hh - 1) * p.W + (C - 1)) * p.xpitch + slot * 8) * 2) : OOB;
        }
    };
    auto dma_x = [&](int i, int cbyte, int buf, int npiece, bool live) {
        const int q = i * 4 + wv;
        const bool go = live && q < npiece;
        const int dst = go ? buf * V10_PB + q * 1024 : V10_DUMP;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_x, (lds_ptr_t)(smem + dst), 16, xsrc[i], cbyte, 0, 0);
    };

    u32x4 Ar[3][4];
    const int a_base = (ct * 4 + wv) * p.nk * 4096;
    const int a_wrap = p.nk * 4096;
    int a_next = 0;
    unsigned a_lane = 0;
    auto a_load = [&](auto SLOT) {
        constexpr int sl = decltype(SLOT)::value;
#pragma unroll
        for (int j = 0; j < 4; ++j) Ar[sl][j] = __builtin_amdgcn_raw_buffer_load_b128(rsrc_wf, a_lane, a_base + a_next + j * 1024, 0);
        a_next += 4096;
        if (a_next == a_wrap) a_next = 0;
    };

    int par = 0;
    unsigned char* const slice = smem + V10_SLICE + wv * (4 * 4096);   // [pass][32 pixels][64 filters]

    // ---- the two accumulator sets and the previous tile's pixel range ([pm0, pmlim): empty before the first tile -- every store and residual load out of range)
    f32x16 acc[MC][MPX], prev[MC][MPX];
#pragma unroll
    for (int a = 0; a < MC; ++a)
#pragma unroll
        for (int b = 0; b < MPX; ++b)
#pragma unroll
            for (int q = 0; q < 16; ++q) prev[a][b][q] = 0.0f;
    int pm0 = 0, pmlim = 0;

    // ---- K loop of one tile of MP column blocks with the previous tile's epilogue inside
    auto k_loop = [&](auto MPC, const int m0, const int m1, const int Qf, const int npiece, const bool has_next, const int nQf, const int nnpiece) {
        constexpr int MP = decltype(MPC)::value;
        int lane = lane0;
        asm volatile("" : "+v"(lane));
        const int frow = lane & 31, fk = lane >> 5;
        a_lane = (unsigned)lane * 16u;
        a_next = 0;
        a_load(IC<0>{});
        a_load(IC<1>{});
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
#pragma unroll
        for (int a = 0; a < MC; ++a)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int cbias = ct * 256 + (wv * MC + a) * 32 + 8 * g + 4 * fk;
                const f32x4 bz = *(const f32x4*)(p.bias + cbias);
#pragma unroll
                for (int b = 0; b < MP; ++b)
#pragma unroll
                    for (int q = 0; q < 4; ++q) acc[a][b][4 * g + q] = bz[q];
            }
        frag B0[MP], B1[MP];
#pragma unroll
        for (int b = 0; b < MP; ++b) B0[b] = *(const frag*)(smem + bb[0][b]);
        int bufd = par ? -V10_PB : V10_PB;

        // state of the pass in flight (one per channel block): its 2 x 16 accumulators, the 8 activated values waiting for their pack, store offsets, residual rows
        f32x16 cur[MC];
        f32x8 sv;
        u32x4 rres[2];        // the residual chunks of the next two packs of the pass (requested two units ahead, in the MFMA layout's own 16-byte pieces)
        int e_frow = 0, e_fk = 0;

        // EPI = 1: the channel block carries pass `hb` of the previous tile; 0: K loop only (channel blocks 4 .. of a tile)
        auto cb_body = [&](auto EPI, const int cb, const int hb) {
            constexpr bool epi = decltype(EPI)::value != 0;
            const bool lastcb = cb + 1 == ncb;
            int np_req = npiece, cbyte = (cb + 1) * 64;
            bool live = true;
            if (lastcb) {
                live = has_next;
                np_req = nnpiece;
                cbyte = 0;
                if (has_next) set_xsrc(nQf - PW - 1, nnpiece);
            }
            const int nbuf = par ^ 1;
            const int m_pass = pm0 + hb * 32;
            static_for<9>([&](auto TAP) {
                constexpr int tap = decltype(TAP)::value;
                constexpr int dh = tap / 3, dw = tap % 3;
                constexpr int ntap = (tap + 1) % 9, ndh = ntap / 3, ndw = ntap % 3;
                // ---- phase 1 (piece 2 tap of the pass: tap 0 = addresses + residual loads, else SiLU unit 2 tap - 1)
#pragma unroll
                for (int b = 0; b < MP; ++b) B1[b] = *(const frag*)(smem + bb[dh][b] + dw * V10_PITCH + 32);
                if constexpr (epi && tap == 0) {
                    int le = lane0;
                    asm volatile("" : "+v"(le));
                    e_frow = le & 31; e_fk = le >> 5;
                }
                a_load(IC<(tap + 2) % 3>{});
                auto unit = [&](auto U) {   // SiLU of two values; every fourth unit packs the eight values of its (filter tile, register half), adds the residual, writes
                    constexpr int u = decltype(U)::value, a = u >> 3, gp = (u >> 2) & 1, pr = u & 3;
                    if constexpr (RES && pr == 1) {   // the 16-byte piece of the residual this lane's pack will cover: pixel e_frow, filters 32 a + 16 gp + 8 fk ..
                        const int m = m_pass + e_frow;
                        const int c = ct * 256 + wv * 64 + (a * 4 + gp * 2 + e_fk) * 8;
                        rres[(u >> 2) & 1] = __builtin_amdgcn_raw_buffer_load_b128(rsrc_r, m < pmlim ? (unsigned)(m * p.rpitch + c) * 2u : OOBS, 0, 0);
                    }
                    f32x2 v = {cur[a][8 * gp + 2 * pr], cur[a][8 * gp + 2 * pr + 1]};
                    silu_vec<f32x2, 2>(v);
                    sv[2 * pr] = v[0];
                    sv[2 * pr + 1] = v[1];
                    if constexpr (pr == 3) {
                        u32x4 ov;
#pragma unroll
                        for (int h = 0; h < 2; ++h) {
                            const auto sw = __builtin_amdgcn_permlane32_swap(pack2<T>(sv[2 * h], sv[2 * h + 1]), pack2<T>(sv[4 + 2 * h], sv[4 + 2 * h + 1]), false, false);
                            ov[h] = (unsigned)sw[0];
                            ov[2 + h] = (unsigned)sw[1];
                        }
                        if constexpr (RES) {   // x + cv2(cv1(x)): fp32 sum of the two stored values, rounded once (epilogue_wave's arithmetic)
                            const vec8 yy = __builtin_bit_cast(vec8, ov), rr = __builtin_bit_cast(vec8, rres[(u >> 2) & 1]);
#pragma unroll
                            for (int q = 0; q < 8; q += 2) ov[q >> 1] = pack2<T>(to_f32<T>(yy[q]) + to_f32<T>(rr[q]), to_f32<T>(yy[q + 1]) + to_f32<T>(rr[q + 1]));
                        }
                        const int chunk = a * 4 + gp * 2 + e_fk;
                        *(u32x4*)(slice + hb * 4096 + e_frow * 128 + ((chunk ^ swz<64>(e_frow)) << 4)) = ov;
                    }
                };
                if constexpr (epi && tap >= 1) unit(IC<2 * tap - 1>{});
#pragma unroll
                for (int a = 0; a < MC; ++a)
#pragma unroll
                    for (int b = 0; b < MP; ++b) acc[a][b] = Mfma<T>::run(__builtin_bit_cast(frag, Ar[tap % 3][a]), B0[b], acc[a][b]);
                {
#pragma unroll
                    for (int i = 0; i < MP; ++i) {
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                        if constexpr (epi) __builtin_amdgcn_sched_group_barrier(0x402, 1, 0);
                    }
                    constexpr int NV1 = MC * MP - MP < 4 ? MC * MP - MP : 4;
#pragma unroll
                    for (int i = 0; i < NV1; ++i) {
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                        __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
                        if constexpr (epi) __builtin_amdgcn_sched_group_barrier(0x402, 1, 0);
                    }
                    if constexpr (4 - NV1 > 0) __builtin_amdgcn_sched_group_barrier(0x020, 4 - NV1, 0);
                    if constexpr (MC * MP - MP - 4 > 0) __builtin_amdgcn_sched_group_barrier(0x008, MC * MP - MP - 4, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
                // ---- phase 2 (SiLU unit 2 tap for taps 0 .. 7)
                if constexpr (tap == 8) {
                    v10_wait_vm<8>();
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    __builtin_amdgcn_s_barrier();
#pragma unroll
                    for (int d = 0; d < 3; ++d)
#pragma unroll
                        for (int b = 0; b < MP; ++b) bb[d][b] += bufd;
                    bufd = -bufd;
                }
#pragma unroll
                for (int b = 0; b < MP; ++b) B0[b] = *(const frag*)(smem + bb[ndh][b] + ndw * V10_PITCH);
                if constexpr (tap < 7) {
#pragma unroll
                    for (int x = 0; x < XQ; ++x) dma_x(tap * XQ + x, cbyte, nbuf, np_req, live);
                }
                if constexpr (epi && tap < 8) unit(IC<2 * tap>{});
#pragma unroll
                for (int a = 0; a < MC; ++a)
#pragma unroll
                    for (int b = 0; b < MP; ++b) acc[a][b] = Mfma<T>::run(__builtin_bit_cast(frag, Ar[tap % 3][2 + a]), B1[b], acc[a][b]);
                {
#pragma unroll
                    for (int i = 0; i < MP; ++i) {
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                        if constexpr (epi) __builtin_amdgcn_sched_group_barrier(0x402, 1, 0);
                    }
                    if constexpr (tap < 7) {
#pragma unroll
                        for (int x = 0; x < XQ; ++x) {
                            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                            __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
                            if constexpr (epi) __builtin_amdgcn_sched_group_barrier(0x402, 1, 0);
                        }
                    }
                    constexpr int REST = MC * MP - MP - (tap < 7 ? XQ : 0);
                    if constexpr (REST > 0) {
                        if constexpr (epi) {
#pragma unroll
                            for (int i = 0; i < REST; ++i) {
                                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                                __builtin_amdgcn_sched_group_barrier(0x402, 1, 0);
                            }
                        } else {
                            __builtin_amdgcn_sched_group_barrier(0x008, REST, 0);
                        }
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            });
            par ^= 1;
        };

        // two loops, one body each: with both bodies under one loop the accumulators of the two paths were allocated apart and met in 128 register moves per trip
        int cb = 0;
        do {   // channel blocks 0 .. 3 carry the four passes of the previous tile (host: Cin >= 128)
            switch (cb) {   // (uniform) the pass's accumulators into `cur`
                case 0:
#pragma unroll
                    for (int a = 0; a < MC; ++a) cur[a] = prev[a][0];
                    break;
                case 1:
#pragma unroll
                    for (int a = 0; a < MC; ++a) cur[a] = prev[a][1];
                    break;
                case 2:
#pragma unroll
                    for (int a = 0; a < MC; ++a) cur[a] = prev[a][2];
                    break;
                default:
#pragma unroll
                    for (int a = 0; a < MC; ++a) cur[a] = prev[a][3];
                    break;
            }
            cb_body(IC<1>{}, cb, cb);
        } while (++cb < MPX);
        if (cb < ncb) {
            do cb_body(IC<0>{}, cb, 0);
            while (++cb < ncb);
        }

        // ---- what is left of the previous tile: its four slices back in store order, the residual, the stores
        {
            int le = lane0;
            asm volatile("" : "+v"(le));
            const int rp = le >> 3, ch = le & 7;
            const int c = ct * 256 + wv * 64 + ch * 8;
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // (the wave's own LDS writes)
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
            for (int h4 = 0; h4 < MPX; ++h4)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int pl = i * 8 + rp;
                    const int m = pm0 + h4 * 32 + pl;
                    vec8 ov = *(const vec8*)(slice + h4 * 4096 + pl * 128 + ((ch ^ swz<64>(pl)) << 4));
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, ov), rsrc_y, m < pmlim ? (unsigned)(m * p.ypitch + c) * 2u : OOBS, 0, 0);
                }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();   // the slices are rewritten by the next tile's passes
        }
    };

    // ---- prologue of the block: the whole patch of (tile 0, channel block 0)
    int m0, m1, Qf, npiece;
    tile_geom(0, m0, m1, Qf, npiece);
    set_xsrc(Qf - PW - 1, npiece);
#pragma unroll
    for (int i = 0; i < NXP; ++i) dma_x(i, 0, 0, npiece, true);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();

    int t = 0;
    do {
        const bool has_next = t + 1 < nt;
        int nm0 = 0, nm1 = 1, nQf = 0, nnp = 0;
        if (has_next) tile_geom(t + 1, nm0, nm1, nQf, nnp);
        const int sz = tq + (t < tr ? 1 : 0);
        if (sz >= 4) k_loop(IC<4>{}, m0, m1, Qf, npiece, has_next, nQf, nnp);
        else k_loop(IC<3>{}, m0, m1, Qf, npiece, has_next, nQf, nnp);
        // the finished tile becomes the one that leaves during the next tile's K loop (a body of three column blocks leaves its fourth pass beyond pmlim)
#pragma unroll
        for (int a = 0; a < MC; ++a)
#pragma unroll
            for (int b = 0; b < MPX; ++b) prev[a][b] = acc[a][b];
        pm0 = m0;
        pmlim = m1 < p.M ? m1 : p.M;
        m0 = nm0; m1 = nm1; Qf = nQf; npiece = nnp;
    } while (++t < nt);

    // ---- the block's last tile: the plain epilogue
    {
        int lane_e = lane0;
        asm volatile("" : "+v"(lane_e));
#pragma unroll
        for (int hb = 0; hb < MPX; ++hb) {
            f32x16 part[MC][1];
#pragma unroll
            for (int a = 0; a < MC; ++a) part[a][0] = prev[a][hb];
            epilogue_wave<T, MC, 1, false, true>(p, part, slice, ct * 256 + wv * MC * 32, pm0 + hb * 32, lane_e, -1, pmlim);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }
    }
#endif
}
