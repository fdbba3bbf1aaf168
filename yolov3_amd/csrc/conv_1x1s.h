// conv_1x1s.h -- included by conv.hip INSIDE its anonymous namespace (shares ConvArgs, Mfma, epilogue_wave, IC / static_for, ...).
//
// "s1x1": the HBM-bound 1x1 / stride-1 convolutions (reference models/common.py:150-165 Bottleneck.cv1 on the 80 x 80 / 160 x 160 maps and their data gradients:
// 256 -> 128, 384 -> 128, 128 -> 64, 128 -> 256, 64 -> 128).  Per pixel they read Cin and write Cout values and multiply almost nothing: the tile kernels run them at
// 4.0-4.3 TB/s (profiles/r05_conv_lab_1x1.txt) because every 128 x 128 tile fetches its filter tile again -- as many bytes through the LDS-DMA path as the pixels
// themselves -- and pays a prologue (first stage exposed) and two barriers per K-step for a K loop of 4-8 steps.  Here:
//   * PERSISTENT blocks (one per CU), the whole filter matrix of the layer in REGISTERS: a consumer wave holds its FG groups of 32 filters for all K (FG x K/16 MFMA A
//     fragments, loaded once per block) and multiplies a stage's pixels once per group;
//   * the pixel operand streams through a ring of three LDS stages laid out in fragment order ([32-pixel column block][16-channel K-step][lane]: 1 KiB per LDS-DMA
//     instruction, a consumer's B fragment is one conflict-free ds_read_b128);
//   * a fifth PRODUCER wave issues every LDS-DMA request of the block and does nothing else, so its vmcnt counts requests only: `s_waitcnt vmcnt(requests of one
//     stage)` = "the stage before the youngest has landed" -- two stages (64 KiB at Cin = 256) in flight per CU at all times, one barrier per stage;
//   * the epilogue is epilogue_wave's (bias in the accumulator, SiLU, residual, statistics rows, channel-slice stores) on passes of two column blocks.
// Geometry per (FG, WC, KS): WC consumer waves along the filters (WC * FG * 32 = Cout), 4 / WC along the pixels; KS = Cin / 16 K-steps; a stage holds as many
// 32-pixel column blocks as make 32 (48 at KS = 24) requests.

// IN (training forward, "consumer-side BatchNorm"): 0 the operand is the tensor as stored; 1 / 2 the operand is the PRE-BatchNorm output u of the producing layer and the kernel
// applies that layer's act(scale u + shift) (2: + its shortcut tensor) on the way in, writes the result y once for the other consumers, and multiplies it -- the producing
// layer's normalise pass and this layer's read of y disappear (reference models/common.py:75, :165).  u streams through the LDS stages like any operand; the shortcut rows do
// not (a stage would hold half as many pixels and the per-stage costs would double): every consumer wave prefetches the shortcut fragments of its quarter of the NEXT stage
// into registers while the current stage is multiplied.
template <int FG_, int WC_, int KS_, int IN_ = 0> struct S1Geom {
    static constexpr int FG = FG_, WC = WC_, KS = KS_, IN = IN_;
    static constexpr int WP = 4 / WC;                       // consumer waves along the pixel axis
    static constexpr int TC = WC * FG * 32;                 // filters per block
    static constexpr int IS = KS == 24 ? 48 : 32;           // LDS-DMA requests (1 KiB each) per stage
    static constexpr int CB = IS / KS;                      // 32-pixel column blocks per stage
    static constexpr int NF = CB * KS;                      // B fragments (1 KiB) of a stage
    static constexpr int SP = CB * 32;                      // pixels per stage
    static constexpr int MP = CB / WP;                      // column blocks per consumer wave and stage
    static constexpr int PB = MP >= 2 ? 2 : 1;              // column blocks per epilogue pass
    static constexpr int NPASS = MP / PB;                   // epilogue passes per stage and wave
    static constexpr int STAGE = IS * 1024;
    static constexpr int NS = 3;
    static constexpr int SLICE_BYTES = PB * 32 * 64;        // one epilogue pass of a wave: PB * 32 pixels x 32 filters
    static constexpr int SLICE = NS * STAGE;
    static constexpr int TAB = SLICE + 4 * SLICE_BYTES;     // IN: the producing layer's (scale, shift) per input channel, fp32
    static constexpr int LDS = TAB + (IN ? 2 * KS * 16 * 4 : 0);
    static_assert(CB * KS == IS && MP * WP == CB && MP >= 1 && MP % PB == 0, "whole column blocks per wave, whole passes");
    static_assert(IN == 0 || NF % 4 == 0, "the transform deals the fragments of a stage to the four consumer waves");
    static_assert(LDS <= 163840, "the LDS of a CU");
};

template <int N> Y3_DEV void s1_wait_vm() { __builtin_amdgcn_s_waitcnt((N & 15) | (7 << 4) | (15 << 8) | ((N >> 4) << 14)); }   // (conv_v10.h::v10_wait_vm)

template <typename T, int FG, int WC, int KS, int IN = 0>
__global__ __launch_bounds__(320, 1) void conv_1x1s_kernel(const ConvArgs p) {
#if defined(__HIP_DEVICE_COMPILE__)
    typedef S1Geom<FG, WC, KS, IN> G;
    typedef typename Mfma<T>::frag frag;
    typedef __attribute__((address_space(3))) void* lds_ptr_t;
    __shared__ __attribute__((aligned(1024))) unsigned char smem[G::LDS];   // the ONLY LDS object

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);   // 0..3 consumers, 4 the producer
    // filter tile ct (n_ct > 1: Cout beyond what four waves keep in registers -- the blocks of one pixel range sit next to each other in the XCD-grouped id order, so the
    // second .. n_ct-th read of a stage's pixels meets the first in that XCD's L2) and slot b of nb: the block's stages are b, b + nb, ...
    const int lid = xcd_remap((int)blockIdx.x, (int)gridDim.x);
    const int nb = (int)gridDim.x / p.n_ct, ct = lid % p.n_ct, b = lid / p.n_ct;
    const int n_stages = p.n_pt;
    const int my = (n_stages - b + nb - 1) / nb;   // (host: nb <= n_stages)

    if (wv == 4) {
        // ---- producer: lane (pixel lane & 31, channel half lane >> 5) of request (column block cb, K-step ks) fetches 8 channels 16 ks + 8 (lane >> 5) .. of pixel
        // 32 cb + (lane & 31) of the stage; the request lands as the 1 KiB the consumers read as ONE B fragment per lane
        const auto rsrc_x = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, (int)p.x_bytes, 0x00020000);
        constexpr unsigned OOB = 0x80000000u;   // stays out of range when the K-step offset is added: the piece lands as zeros
        const int pl = lane & 31, fk = lane >> 5;
        auto issue = [&](int j) {
            const int m0 = (b + j * nb) * G::SP;
            unsigned char* dst = smem + (j % G::NS) * G::STAGE;
#pragma unroll
            for (int cb = 0; cb < G::CB; ++cb) {
                const int m = m0 + cb * 32 + pl;
                const unsigned vo = (j < my && m < p.M) ? (unsigned)((m * p.xpitch + fk * 8) * 2) : OOB;
#pragma unroll
                for (int ks = 0; ks < KS; ++ks)
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_x, (lds_ptr_t)(dst + (cb * KS + ks) * 1024), 16, vo, ks * 32, 0, 0);
            }
        };
        issue(0);
        issue(1);
        for (int j = 0; j < my; ++j) {
            s1_wait_vm<G::IS>();              // only the requests of stage j + 1 are younger than stage j's: stage j has landed
            __builtin_amdgcn_s_barrier();     // ... and is published; every consumer is done with stage j - 1, whose slot stage j + 2 takes
            issue(j + 2);                     // (beyond the block's last stage: out-of-range requests, so that the count above stays exact)
            if constexpr (IN != 0) __builtin_amdgcn_s_barrier();   // (the consumers' barrier between transforming the stage and multiplying it)
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // nothing of this wave is in flight into an LDS that the next workgroup may own
        return;
    }

    // ---- consumers
    const int wc = wv / G::WP, wp = wv % G::WP;
    const int frow = lane & 31, fk = lane >> 5;
    const auto rsrc_w = __builtin_amdgcn_make_buffer_rsrc((void*)p.w, 0, (int)p.w_bytes, 0x00020000);
    frag Wr[FG][KS];   // the wave's filters for all K: row (wc FG + fg) 32 + frow, channels 16 ks + 8 fk ..
#pragma unroll
    for (int fg = 0; fg < FG; ++fg)
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const unsigned off = (unsigned)(((ct * G::TC + (wc * FG + fg) * 32 + frow) * p.Kpad + ks * 16 + fk * 8) * 2);
            Wr[fg][ks] = __builtin_bit_cast(frag, __builtin_amdgcn_raw_buffer_load_b128(rsrc_w, off, 0, 0));
        }
    f32x4 bz[FG][4];   // the accumulators start at the bias of their filter (lane holds filters 8 g + 4 fk + q of each 32-filter group)
#pragma unroll
    for (int fg = 0; fg < FG; ++fg)
#pragma unroll
        for (int g = 0; g < 4; ++g) bz[fg][g] = *(const f32x4*)(p.bias + ct * G::TC + (wc * FG + fg) * 32 + 8 * g + 4 * fk);
    unsigned char* slice = smem + G::SLICE + wv * G::SLICE_BYTES;

    // IN: the producing layer's (scale, shift) table in LDS (published by the first stage's barrier), the shortcut fragments of this wave's quarter of a stage in registers
    const auto rsrc_iy = __builtin_amdgcn_make_buffer_rsrc((void*)(IN ? p.in_y : p.y), 0, IN ? (int)p.in_y_bytes : 0, 0x00020000);
    const auto rsrc_ir = __builtin_amdgcn_make_buffer_rsrc((void*)(IN == 2 ? p.in_res : p.y), 0, IN == 2 ? (int)p.in_r_bytes : 0, 0x00020000);
    constexpr int NQ = IN ? G::NF / 4 : 1;   // fragments of a stage per consumer wave: f = wv + 4 i
    u32x4 rpre[NQ];
    auto prefetch_shortcut = [&](int j) {
        if constexpr (IN == 2) {
            const int m0 = (b + j * nb) * G::SP;
#pragma unroll
            for (int i = 0; i < NQ; ++i) {
                const int f = wv + 4 * i, cb = f / KS, ks = f - cb * KS;
                const int m = m0 + cb * 32 + frow;
                rpre[i] = __builtin_amdgcn_raw_buffer_load_b128(rsrc_ir, (j < my && m < p.M) ? (unsigned)((m * p.in_rpitch + ks * 16 + fk * 8) * 2) : 0x80000000u, 0, 0);
            }
        }
    };
    if constexpr (IN != 0) {
        float* tab = (float*)(smem + G::TAB);
        for (int c = wv * 64 + lane; c < KS * 16; c += 256) { tab[c] = p.in_scale[c]; tab[KS * 16 + c] = p.in_shift[c]; }
        prefetch_shortcut(0);
    }

    for (int j = 0; j < my; ++j) {
        if constexpr (IN != 0) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // (first trip: this wave's part of the table is written)
        __builtin_amdgcn_s_barrier();   // stage j has landed for every wave
        const int s = b + j * nb;
        if constexpr (IN != 0) {
            // y = act(scale u + shift) (+ shortcut), the arithmetic of bn_act_fwd_kernel (train.hip: y3_bn_act2): each consumer wave transforms a quarter of the stage's
            // fragments IN PLACE and stores them -- a lane's fragment is 8 consecutive channels of one pixel: a 16-byte piece of y's row
            unsigned char* su = smem + (j % G::NS) * G::STAGE + lane * 16;
            const float* tab = (const float*)(smem + G::TAB);
#pragma unroll
            for (int i = 0; i < NQ; ++i) {
                const int f = wv + 4 * i;                   // (wave-uniform)
                const int cb = f / KS, ks = f - cb * KS;
                typedef typename Mfma<T>::frag vec8;
                const vec8 uu = *(const vec8*)(su + f * 1024);
                const int c0 = ks * 16 + fk * 8;
                const f32x4 sc0 = *(const f32x4*)(tab + c0), sc1 = *(const f32x4*)(tab + c0 + 4), sh0 = *(const f32x4*)(tab + KS * 16 + c0), sh1 = *(const f32x4*)(tab + KS * 16 + c0 + 4);
                const float scv[8] = {sc0[0], sc0[1], sc0[2], sc0[3], sc1[0], sc1[1], sc1[2], sc1[3]};
                const float shv[8] = {sh0[0], sh0[1], sh0[2], sh0[3], sh1[0], sh1[1], sh1[2], sh1[3]};
                vec8 rr;
                if constexpr (IN == 2) rr = __builtin_bit_cast(vec8, rpre[i]);
                u32x4 ov;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    f32x2 r2 = {0.0f, 0.0f};
                    if constexpr (IN == 2) r2 = f32x2{to_f32<T>(rr[2 * q]), to_f32<T>(rr[2 * q + 1])};
                    const f32x2 z = y3_bn_act2(f32x2{to_f32<T>(uu[2 * q]), to_f32<T>(uu[2 * q + 1])}, f32x2{scv[2 * q], scv[2 * q + 1]}, f32x2{shv[2 * q], shv[2 * q + 1]},
                                               p.in_act == Y3_ACT_SILU, IN == 2, r2);
                    ov[q] = pack2<T>(z[0], z[1]);
                }
                *(u32x4*)(su + f * 1024) = ov;
                const int m = s * G::SP + cb * 32 + frow;
                __builtin_amdgcn_raw_buffer_store_b128(ov, rsrc_iy, m < p.M ? (unsigned)((m * p.in_ypitch + c0) * 2) : 0x80000000u, 0, 0);
            }
            prefetch_shortcut(j + 1);       // in flight under this stage's multiplication
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // this wave's fragments are written (LDS only: the y stores and the prefetch stay in flight)
            __builtin_amdgcn_s_barrier();   // every fragment of the stage is y now
        }
        const unsigned char* st = smem + (j % G::NS) * G::STAGE + (wp * G::MP) * KS * 1024 + lane * 16;
        const int m_wave = s * G::SP + wp * G::MP * 32;
        // one epilogue pass (PB column blocks) and one filter group at a time: multiply, then hand the 32 x PB * 32 tile to the epilogue (16 PB accumulator registers live)
        static_for<G::NPASS * FG>([&](auto IT) {
            constexpr int hb = decltype(IT)::value / FG, fg = decltype(IT)::value % FG;
            f32x16 acc[1][G::PB];
#pragma unroll
            for (int cb = 0; cb < G::PB; ++cb)
#pragma unroll
                for (int g = 0; g < 4; ++g)
#pragma unroll
                    for (int q = 0; q < 4; ++q) acc[0][cb][4 * g + q] = bz[fg][g][q];
#pragma unroll
            for (int cb = 0; cb < G::PB; ++cb)
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
                    const frag B = *(const frag*)(st + ((G::PB * hb + cb) * KS + ks) * 1024);
                    acc[0][cb] = Mfma<T>::run(Wr[fg][ks], B, acc[0][cb]);
                }
            epilogue_wave<T, 1, G::PB, false, true>(p, acc, slice, ct * G::TC + (wc * FG + fg) * 32, m_wave + hb * G::PB * 32, lane, (s * G::WP + wp) * G::NPASS + hb, p.M);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();   // the slice is private to the wave: the reads of one pass precede the writes of the next
        });
    }
#endif
}

// the (filter groups per wave, waves along the filters, K-steps) form of a layer, or false: not one of the HBM-bound 1x1 shapes
struct S1Plan {
    int fg, wc, ks, sp, wp, npass, n_ct, in;
};
static bool s1_plan(const ConvArgs& a, S1Plan& pl, int in = 0) {   // in: 0 plain operand; 1 / 2 the producing layer's BatchNorm + activation (+ shortcut) applied on the way in
    const long long mode = y3_knob(Y3K_CONV_1X1S);
    if (mode == 0 || a.ups || a.dil_shift) return false;
    if (a.ks != 1 || a.ntaps != 1 || a.stride != 1 || a.pad != 0 || a.tdh[0] != 0 || a.tdw[0] != 0) return false;
    if (a.omul != 1 || a.ooh != 0 || a.oow != 0 || a.H != a.Ho || a.W != a.Wo || a.oH != a.Ho || a.oW != a.Wo) return false;
    if (!a.x_bytes || !a.w_bytes || !a.y_bytes || (a.res && !a.r_bytes) || !a.bias) return false;
    if (a.M < 32768 && mode != 2 && !in) return false;   // below that the launch is latency-bound and the tile kernels' many small blocks win
    // the IN form (consumer-side BatchNorm) replaces TWO launches, so it pays earlier -- but not at any size: with a handful of stages per persistent block (a short last
    // batch, the 320-pixel end of multi-scale training) it is the one-block-per-CU regime the tile kernels were measured faster in; the caller keeps the separate passes
    // (y3_conv2d_fwd_bnin_rows returns -1).  The committed A/Bs (profiles/r05_bn_in_pairs_probe.txt) are at >= 102 400 pixels.
    if (a.M < 8192 && mode != 2 && in) return false;
    // WC consumer waves along the filters x FG groups of 32 filters per wave = the filters of a block; n_ct blocks side by side where Cout is more than a block keeps in
    // registers.  The shapes of yolov3's Bottleneck.cv1 layers on the 40 x 40 ... 320 x 320 maps, their data gradients, the 80 x 80 Detect conv (255 -> 256 filters) and the
    // cv1 layers behind the two Concats.  Cin = 512 (the 40 x 40 cv1 layers, 128 filters per block and two blocks per pixel range) was built and measured: 29.1 us against
    // v6's 27.1 at batch 32, level at batch 64 (profiles/r05_conv_lab_s1x1_c.txt) -- those launches stay on v6; 768 -> 256 does not fit (48 K-steps)
    int fg, wc, n_ct = 1;
    const int k = a.Cin / 16;
    if ((a.Cin % 16) || k < 2) return false;
    if (k == 16 && a.Cout > 256) {            // Cin = 256, Cout = 512 / 768 (data gradients of the 40 x 40 cv1 layers): 256 filters per block
        if (a.Cout % 256) return false;
        fg = 2; wc = 4; n_ct = a.Cout / 256;
    } else {
        switch (a.Cout) {
            case 32: fg = 1; wc = 1; break;
            case 64: fg = 1; wc = 2; break;
            case 128: fg = 1; wc = 4; break;
            case 256: fg = 2; wc = 4; break;
            case 384: fg = 3; wc = 4; break;
            default: return false;
        }
    }
    if (n_ct > 8) return false;
    {   // the instantiations that exist (launch_s1)
        const bool ok = (fg == 1 && wc == 4 && (k == 4 || k == 16 || k == 24)) || (fg == 1 && wc == 2 && (k == 2 || k == 8)) || (fg == 1 && wc == 1 && k == 4) ||
                        (fg == 2 && wc == 4 && (k == 8 || k == 16)) || (fg == 3 && wc == 4 && k == 8);
        if (!ok) return false;
    }
    if (in) {   // the forward cv1 shapes of the Bottlenecks: 128 -> 64, 256 -> 128 (64 -> 32 @320x320 behind layer 1 was measured 1.38 -> 1.40 ms for the pair: not covered,
                // profiles/r05_bn_in_pairs_probe.txt)
        const bool ok = n_ct == 1 && fg == 1 && ((wc == 4 && k == 16) || (wc == 2 && k == 8));
        if (!ok || !a.in_scale || !a.in_shift || !a.in_y || !a.in_y_bytes || (in == 2 && (!a.in_res || !a.in_r_bytes))) return false;
    }
    const int ks = k;
    const int is = ks == 24 ? 48 : 32;
    const int cb = is / ks, wp = 4 / wc, mp = cb / wp;
    if (mp < 1) return false;
    pl.fg = fg; pl.wc = wc; pl.ks = ks; pl.n_ct = n_ct; pl.in = in;
    pl.sp = cb * 32;
    pl.wp = wp;
    pl.npass = mp >= 2 ? mp / 2 : 1;
    return true;
}

template <typename T> int launch_s1(ConvArgs& a, const S1Plan& pl, hipStream_t st) {
    a.n_ct = pl.n_ct;
    a.n_pt = y3_ceil_div(a.M, pl.sp);          // stages
    a.stat_wp = pl.wp * pl.npass;              // statistics rows per stage: one per (pixel wave, epilogue pass)
    set_divisors(a);
    g_last_variant = pl.in ? "s1x1_bn" : "s1x1";
    if (a.dry) return 0;
    const int cus = y3_cu_count();
    int slots = cus / pl.n_ct;                 // blocks per filter tile
    if (slots < 1) slots = 1;
    if (slots > a.n_pt) slots = a.n_pt;
    const dim3 grid((unsigned)(slots * pl.n_ct)), block(320);
#define Y3_S1_CASE(FG, WC, KS) if (pl.in == 0 && pl.fg == FG && pl.wc == WC && pl.ks == KS) hipLaunchKernelGGL((conv_1x1s_kernel<T, FG, WC, KS>), grid, block, 0, st, a)
#define Y3_S1_IN(FG, WC, KS, IN) if (pl.in == IN && pl.fg == FG && pl.wc == WC && pl.ks == KS) hipLaunchKernelGGL((conv_1x1s_kernel<T, FG, WC, KS, IN>), grid, block, 0, st, a)
    Y3_S1_CASE(1, 4, 16);        // 256 -> 128
    else Y3_S1_CASE(1, 4, 24);   // 384 -> 128
    else Y3_S1_CASE(1, 4, 4);    //  64 -> 128
    else Y3_S1_CASE(1, 2, 8);    // 128 ->  64
    else Y3_S1_CASE(1, 2, 2);    //  32 ->  64
    else Y3_S1_CASE(1, 1, 4);    //  64 ->  32
    else Y3_S1_CASE(2, 4, 8);    // 128 -> 256
    else Y3_S1_CASE(2, 4, 16);   // 256 -> 256 n_ct (the 80 x 80 Detect conv and its data gradient; 256 -> 512: data gradients of the 40 x 40 cv1 layers)
    else Y3_S1_CASE(3, 4, 8);    // 128 -> 384
    else Y3_S1_IN(1, 4, 16, 1);  // the same cv1 shapes behind a layer without / with a shortcut, that layer's BatchNorm + activation applied on the way in
    else Y3_S1_IN(1, 4, 16, 2);
    else Y3_S1_IN(1, 2, 8, 1);
    else Y3_S1_IN(1, 2, 8, 2);
#undef Y3_S1_IN
#undef Y3_S1_CASE
    else Y3_FAIL("conv s1x1: no instantiation (internal)");
    Y3_CHECK_LAUNCH();
    return 0;
}
