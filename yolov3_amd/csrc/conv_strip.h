// conv_strip.h -- included by conv.hip INSIDE its anonymous namespace, after conv_v10.h (shares ConvArgs, Mfma, epilogue_wave, fdiv, ...).
//
// 3x3 / pad 1 convolutions (stride 1, and 64 -> 128 at stride 2: layer 3) with FEW channels on LARGE maps (reference models/yolov3.yaml:17-22, Bottleneck.cv2 of layers 2 and 4 in training
// mode, models/common.py:57-81,150-165, and their data gradients, which are the same convolution with the channel counts swapped):
// 64 -> 32, 64 -> 128 and 128 -> 64 channels at 320x320 / 160x160.
//
// Why (profiles/r03_wgrad_strip_ab.txt and the staged-byte accounting in csrc/wgrad_strip.h): the tile kernels stage the pixel operand once per TAP --
// 9 x 64-128 B of `buffer_load ... lds` per pixel and filter tile for 64-128 B that exist -- and on these layers that request stream (10-13 TB/s of
// staged bytes), not HBM and not the matrix pipe, is the bound: 32 -> 64 @320x320 runs at 0.41, its data gradient at 0.35 PFLOP/s.  Here:
//   * the filters live in REGISTERS: a wave owns 32 filters and 36 of the 9 Cin / 16 reduction steps (36 MFMA A-fragments, 144 VGPRs), loaded once per
//     block -- no filter staging, no filter fragment reads.  Cin = 128 has 72 steps: two waves share a (pixel tile, filter tile) and split them (KS = 2);
//     after the MFMAs of an output row one of the two hands its 32 x 32 partial sums over through LDS and the other adds them and runs the epilogue, the
//     roles alternating from row to row so that both do the same work;
//   * a block walks down a column strip of MT x 32 pixels; an output row needs three input rows, which live in a ring of four row buffers -- ONE new input
//     row is requested per output row (one K-step ahead) and the nine taps are nine views of the three resident rows (row buffer = kh, pixel shift = kw:
//     instruction immediates).  Padding is free: out-of-image rows / columns are lanes with an out-of-range source offset (the descriptor lands zeros);
//   * pixel rows have a pitch of (2 Cin + 16) bytes: the 32 lanes of a fragment read (consecutive pixels, 16 bytes each) fall on distinct bank quads, the
//     conflict-free property conv_v10.h buys the same way (every 9th / 17th 16-byte slot of a request is a pad slot: an out-of-range lane);
//   * waves = MT pixel tiles x Cout / 32 filter tiles (x KS); per output row a wave issues 36 ds_read_b128 + as many MFMAs and one epilogue_wave call
//     (bias, SiLU, NHWC transpose through its private LDS slice, 16-byte stores); BatchNorm statistics accumulate in 16 registers over the block's rows and
//     leave as ONE row per wave (epilogue_stats_flush) instead of one per 64 pixels;
//   * persistent blocks over the linear index t = ((image, strip), row), like wgrad_strip.h.
// Measured at batch 64 (profiles/r03_conv_strip_ab.txt): 64 -> 32 @320x320 773 -> 314 us, 128 -> 64 @160x160 544 -> 342 us, 64 -> 128 @160x160 335 -> 314 us;
// the KS = 1 forms are bit-identical to the tile kernels (the same products added in the same order).  32 -> 64 itself (18 steps per epilogue call)
// measured level with the 64 x 256-pixel tile kernel and stays there.

template <int CIN, int COUT, int MT, int S = 1> struct CsGeom {
    static constexpr int KS = CIN > 64 ? 2 : 1;                          // waves that split the reduction of one (pixel tile, filter tile)
    static constexpr int PXB = CIN * 2, PP = PXB + 16, SLOTS = PP / 16;   // pixel bytes, pitch, 16-byte slots per pixel (the last one is the pad)
    static constexpr int SWP = MT * 32, NPX = S * SWP + (S == 1 ? 2 : 1); // strip width (output pixels), input pixels of a row segment
    // stride 2: output pixel m reads input pixels 2 m + kw -- the even and the odd input pixels of a row sit in two halves of the row buffer, so that the 32
    // lanes of a fragment read are consecutive pixels of one half (kw = 0 / 2: even half at m / m + 1, kw = 1: odd half at m)
    static constexpr int NEVEN = S == 1 ? NPX : SWP + 1;                  // pixels of the first half
    static constexpr int XPIECES = (NPX * PP + 1023) / 1024, XROWB = XPIECES * 1024;
    static constexpr int NSLOT = S == 1 ? 4 : 5;                          // three rows in use + the S being fetched
    static constexpr int NT = COUT / 32, NW = MT * NT * KS;
    static constexpr int KPT = CIN / 16, KF = 9 * KPT / KS;               // 16-wide reduction steps per tap / per wave
    static constexpr int EPI = NW * 2048;                                 // the waves' transpose slices (32 pixels x 64 bytes)
    static constexpr int PART = KS > 1 ? MT * NT * 4096 : 0;              // hand-over buffers of the K-split pairs (16 fp32 per lane)
    static constexpr int XBASE = EPI + PART, LDS = XBASE + NSLOT * XROWB;
    static_assert(KF <= 36 && (9 * KPT) % KS == 0, "144 filter registers per lane");
    static constexpr int XPW = (XPIECES + NW - 1) / NW;
};
template <int CIN, int COUT, int MT> constexpr int cs_threads() { return CsGeom<CIN, COUT, MT>::NW * 64; }   // (the same for both strides)

template <typename T, int CIN, int COUT, int MT, int S = 1>
__global__ __launch_bounds__((cs_threads<CIN, COUT, MT>())) void conv_strip_kernel(const ConvArgs p) {
#if defined(__HIP_DEVICE_COMPILE__)
    typedef CsGeom<CIN, COUT, MT, S> G;
    constexpr int PP = G::PP, XROWB = G::XROWB, NW = G::NW, NT = G::NT, KF = G::KF, KPT = G::KPT, XPW = G::XPW, KS = G::KS;
    typedef typename Mfma<T>::frag frag;
    typedef __attribute__((address_space(3))) void* lds_ptr_t;
    __shared__ __attribute__((aligned(1024))) unsigned char smem[G::LDS];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ks = wv % KS, pair = wv / KS;           // (K-split index, (pixel tile, filter tile) pair)
    const int mt = pair / NT, nt = pair % NT;
    const int frow = lane & 31, fk = lane >> 5;
    const int t_begin = blockIdx.x * p.cs_per;
    int t_end = t_begin + p.cs_per;
    if (t_end > p.cs_T) t_end = p.cs_T;
    if (t_begin >= t_end) return;
    const auto rsrc_x = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, (int)p.x_bytes, 0x00020000);
    constexpr unsigned OOB = 0xffffffffu;

    // ---- the wave's filters: rows nt 32 + frow of the packed bank, all 9 Cin reduction steps (lane: k-group fk of every 16-step)
    frag fw[KF];
    {
        const T* wrow = (const T*)p.w + (size_t)(nt * 32 + frow) * p.Kpad + ks * (KF * 16) + fk * 8;
#pragma unroll
        for (int j = 0; j < KF; ++j) fw[j] = *(const frag*)(wrow + 16 * j);
    }
    f32x4 bz[4];   // the accumulators of every output row start at the bias of the lane's filters (8 g + 4 fk + q)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const int cb = nt * 32 + 8 * g + 4 * fk;
        bz[g] = (p.bias && ks == 0) ? *(const f32x4*)(p.bias + cb) : f32x4{0.f, 0.f, 0.f, 0.f};   // (K-split: the bias enters once)
    }

    // ---- staging role: request q = i NW + wave of a row buffer, lane -> LDS position q 1024 + 16 lane = (pixel e / SLOTS, slot e % SLOTS), e = q 64 + lane
    int xj[XPW];
    int xoffl[XPW];
#pragma unroll
    for (int i = 0; i < XPW; ++i) {
        const int e = (i * NW + wv) * 64 + lane;
        const int pos = e / G::SLOTS, slot = e - pos * G::SLOTS;   // position in the row buffer
        const int j = S == 1 ? pos : (pos < G::NEVEN ? 2 * pos : 2 * (pos - G::NEVEN) + 1);   // input pixel of the segment (stride 2: even half, then odd half)
        xj[i] = (slot < G::SLOTS - 1 && pos < G::NPX) ? j : -1;    // pad slots and the tail of the last request fetch nothing
        xoffl[i] = ((j - 1) * p.xpitch + slot * 8) * 2;            // relative to input pixel (row, w0 S)
    }
    // ---- fragment address of the wave's pixel tile: pixel mt 32 + frow of the strip, k-group fk; (row buffer, kw, 16-step of the tap) are added per read
    const unsigned char* xfrag = smem + G::XBASE + (mt * 32 + frow) * PP + fk * 16;
    unsigned char* slice = smem + wv * 2048;

    float sacc[16];
#pragma unroll
    for (int q = 0; q < 16; ++q) sacc[q] = 0.0f;

    // ---- the walk (see wgrad_strip.h): (image, strip, row) of the output row requested next, ring slots of the three input rows of the current / next row
    int tn = fdiv(t_begin, p.dv_pw_mul, p.dv_pw_sh);          // host: reciprocal of Ho
    int row = t_begin - tn * p.Ho;
    int img = fdiv(tn, p.dv_h1_mul, p.dv_h1_sh);              // host: reciprocal of the strips per row
    int strip = tn - img * p.cs_strips;
    int cur[3] = {0, 0, 0}, nxt[3] = {0, 0, 0};
    int hp = 0;
    int c_img = 0, c_strip = 0, c_row = 0;                    // the output row being computed (one behind the request cursor)
    int n_img = 0, n_strip = 0, n_row = 0;
    auto take = [&]() { const int s = hp; hp = hp + 1 == G::NSLOT ? 0 : hp + 1; return s; };

    auto request_row = [&](int hin, int slot, int w0) {   // w0: first INPUT column under the strip (output column x stride)
        const bool rowok = (unsigned)hin < (unsigned)p.H;
        const long long xbase = (((long long)img * p.H + hin) * p.W + w0) * p.xpitch * 2;
#pragma unroll
        for (int i = 0; i < XPW; ++i) {
            if ((i + 1) * NW <= G::XPIECES || i * NW + wv < G::XPIECES) {
                const int col = w0 - 1 + xj[i];
                const bool ok = rowok && xj[i] >= 0 && (unsigned)col < (unsigned)p.W;
                const unsigned off = ok ? (unsigned)(xbase + xoffl[i]) : OOB;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_x, (lds_ptr_t)(smem + G::XBASE + slot * XROWB + (i * NW + wv) * 1024), 16, off, 0, 0, 0);
            }
        }
    };
    // requests of output row t (the cursor's): a fresh strip needs three rows (only ever called when no row buffer is in use), a continued one the new bottom row
    auto issue = [&](bool fresh) {
        const int w0 = strip * G::SWP * S;
        if (fresh) {
            nxt[0] = take(); nxt[1] = take(); nxt[2] = take();
            request_row(row * S - 1, nxt[0], w0);
            request_row(row * S, nxt[1], w0);
            request_row(row * S + 1, nxt[2], w0);
        } else if (S == 1) {
            nxt[0] = nxt[1]; nxt[1] = nxt[2]; nxt[2] = take();
            request_row(row + 1, nxt[2], w0);
        } else {   // stride 2: the bottom row of the previous output row is the top row of this one
            nxt[0] = nxt[2]; nxt[1] = take(); nxt[2] = take();
            request_row(row * 2, nxt[1], w0);
            request_row(row * 2 + 1, nxt[2], w0);
        }
        n_img = img; n_strip = strip; n_row = row;
        if (++row == p.Ho) {
            row = 0;
            if (++strip == p.cs_strips) { strip = 0; ++img; }
        }
    };

    auto compute = [&](auto KSC, int t) {
        constexpr int KSI = decltype(KSC)::value;   // the wave's K-split index: its reduction steps are KSI KF .. KSI KF + KF - 1
        f32x16 acc[1][1];
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int q = 0; q < 4; ++q) acc[0][0][4 * g + q] = bz[g][q];
        const unsigned char* xr[3] = {xfrag + cur[0] * XROWB, xfrag + cur[1] * XROWB, xfrag + cur[2] * XROWB};
        // fragment reads run D 16-steps ahead of the MFMAs that consume them (a window of D fragments: the filters hold most of the registers)
        constexpr int D = 6;
        frag xf[D];
        auto ld = [&](auto J) -> frag {
            constexpr int j = KSI * KF + decltype(J)::value;
            constexpr int tap = j / KPT, kh = tap / 3, kw = tap % 3, sub = j % KPT;
            constexpr int px = S == 1 ? kw : (kw == 1 ? G::NEVEN : kw / 2);   // buffer position of the tap relative to the lane's pixel
            return *(const frag*)(xr[kh] + px * PP + sub * 32);
        };
        static_for<D>([&](auto J) { xf[decltype(J)::value] = ld(J); });
        static_for<KF>([&](auto J) {
            constexpr int j = decltype(J)::value;
            acc[0][0] = Mfma<T>::run(fw[j], xf[j % D], acc[0][0]);
            if constexpr (j + D < KF) xf[j % D] = ld(IC<j + D>{});
        });
        if constexpr (KS > 1) {
            // the pair's two partial sums meet in LDS: the giver of this row stores its 16 registers, the taker adds them (roles alternate row by row)
            f32x4* part = (f32x4*)(smem + G::EPI + pair * 4096);
            const bool giver = ((t ^ KSI) & 1) != 0;
            if (giver) {
#pragma unroll
                for (int g = 0; g < 4; ++g) part[g * 64 + lane] = f32x4{acc[0][0][4 * g], acc[0][0][4 * g + 1], acc[0][0][4 * g + 2], acc[0][0][4 * g + 3]};
            }
            __builtin_amdgcn_s_barrier();
            if (giver) return;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4 o = part[g * 64 + lane];
#pragma unroll
                for (int q = 0; q < 4; ++q) acc[0][0][4 * g + q] += o[q];
            }
        }
        const int m_row = (c_img * p.Ho + c_row) * p.Wo;
        const int m_base = m_row + c_strip * G::SWP + mt * 32;
        epilogue_wave<T, 1, 1, true>(p, acc, slice, nt * 32, m_base, lane, -1, m_row + p.Wo, sacc);
    };

    // one output row per iteration: its input rows have landed (requested an iteration ago); the next row's new input row is requested before the MFMAs.
    // A row that starts a new strip finds every row buffer busy or stale: its three rows are requested after the current row's reads (a second barrier), rare.
    issue(true);
    for (int t = t_begin; t < t_end; ++t) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();   // row t's input rows are in LDS for every wave; nobody reads the buffers of row t - 1 any more
        cur[0] = nxt[0]; cur[1] = nxt[1]; cur[2] = nxt[2];
        c_img = n_img; c_strip = n_strip; c_row = n_row;
        const bool more = t + 1 < t_end;
        const bool fresh = more && row == 0;
        if (more && !fresh) issue(false);
        if constexpr (KS == 1) compute(IC<0>{}, t);
        else if (ks == 0) compute(IC<0>{}, t);
        else compute(IC<1>{}, t);
        if (fresh) {
            __builtin_amdgcn_s_barrier();   // every wave is done with this strip's row buffers
            issue(true);
        }
    }
    if (p.stats) epilogue_stats_flush<1>(p, sacc, nt * 32, lane, (blockIdx.x * MT + mt) * KS + ks);
#endif
}

struct CsPlan {
    int blocks, per, strips, T, mt;
};
// knob "conv_strip": 1 on, 0 off, 2 also small launches, N > 2: N output rows per block (tests)
static bool cs_plan(const ConvArgs& a, CsPlan& pl) {
    const long long mode = y3_knob(Y3K_CONV_STRIP);
    if (mode == 0 || a.ups || a.res) return false;
    if (a.ks != 3 || (a.stride != 1 && a.stride != 2) || a.pad != 1 || a.dil_shift != 0 || a.ntaps != 9 || a.omul != 1 || a.ooh != 0 || a.oow != 0) return false;
    if (a.Ho != (a.H - 1) / a.stride + 1 || a.Wo != (a.W - 1) / a.stride + 1 || a.oH != a.Ho || a.oW != a.Wo) return false;
    if (a.stride == 1 ? !((a.Cin == 64 && a.Cout == 32) || (a.Cin == 64 && a.Cout == 128) || (a.Cin == 128 && a.Cout == 64)) : !(a.Cin == 64 && a.Cout == 128)) return false;
    if (!a.x_bytes || !a.w_bytes || !a.y_bytes) return false;
    for (int t = 0; t < 9; ++t)
        if (a.tdh[t] != t / 3 || a.tdw[t] != t % 3) return false;
    pl.mt = 2;
    const int swp = pl.mt * 32;
    pl.strips = (a.Wo + swp - 1) / swp;
    const long long T = (long long)a.N * pl.strips * a.Ho;
    if (T < 1 || T > 0x3fffffffLL) return false;
    // blocks a CU holds (CsGeom::LDS = 44 / 56 / 104 KiB for 64 -> 32 / 64 -> 128 / 128 -> 64; 2 / 8 / 8 waves per block, two waves per SIMD: 144 filter
    // registers per lane)
    const int per_cu = a.Cout == 32 ? 3 : 1;
    const int nblk = y3_cu_count() * per_cu;
    if (mode == 1 && T < 24LL * nblk) return false;   // every block amortises its filter load and its statistics row over >= 24 output rows
    pl.per = mode > 2 ? (int)mode : (int)((T + nblk - 1) / nblk);
    if (pl.per < 1) pl.per = 1;
    pl.blocks = (int)((T + pl.per - 1) / pl.per);
    pl.T = (int)T;
    return true;
}

template <typename T> int launch_cs(ConvArgs& a, hipStream_t st) {
    CsPlan pl;
    if (!cs_plan(a, pl)) Y3_FAIL("conv strip: no plan (internal)");
    a.cs_strips = pl.strips; a.cs_T = pl.T; a.cs_per = pl.per;
    set_divisors(a);
    magic_u31(a.Ho, a.dv_pw_mul, a.dv_pw_sh);          // this kernel divides the row index by Ho and by the strips per row
    magic_u31(pl.strips, a.dv_h1_mul, a.dv_h1_sh);
    a.n_pt = pl.blocks;
    a.n_ct = 1;
    a.stat_wp = pl.mt * (a.Cin > 64 ? 2 : 1);   // one statistics row per block, pixel tile and K-split wave
    g_last_variant = "strip";
    if (a.dry) return 0;
    const dim3 grid((unsigned)pl.blocks);
    if (a.stride == 2) hipLaunchKernelGGL((conv_strip_kernel<T, 64, 128, 2, 2>), grid, dim3(cs_threads<64, 128, 2>()), 0, st, a);
    else if (a.Cin == 128) hipLaunchKernelGGL((conv_strip_kernel<T, 128, 64, 2>), grid, dim3(cs_threads<128, 64, 2>()), 0, st, a);
    else if (a.Cout == 32) hipLaunchKernelGGL((conv_strip_kernel<T, 64, 32, 2>), grid, dim3(cs_threads<64, 32, 2>()), 0, st, a);
    else hipLaunchKernelGGL((conv_strip_kernel<T, 64, 128, 2>), grid, dim3(cs_threads<64, 128, 2>()), 0, st, a);
    Y3_CHECK_LAUNCH();
    return 0;
}

// ---- the stride-2 data gradient on the same plan ---------------------------------------------------------------------------------------------------
// dx of a 3x3 / stride-2 convolution (reference models/yolov3.yaml:16,19: layers 1 and 3; y3_conv2d_dgrad_s2) = four stride-1 convolutions of du, one per
// output-pixel parity (ph, pw), with 1 / 2 / 2 / 4 taps whose input offsets are (dh, dw) in {0, 1}^2 (S2Class in conv.hip): 9 (tap, class) pairs over FOUR
// distinct shifted views of du.  The tile kernels (conv_igemm_v3_quad_kernel) stage a du tile once per tap and class -- 9 x 128-256 B per du pixel.  Here a
// K-step is one du row of a 64-pixel strip: rows i and i + 1 are resident (ring of three, one new row per step), a wave reads each of the 4 x Cin / 16
// shifted fragments ONCE and feeds it to every (tap, class) that uses it -- back-to-back MFMAs into up to four independent accumulators -- and the filters
// of its classes sit in registers: 64 -> 32 channels (layer 1 of yolov3, the largest data gradient of the step) all four classes, 36 fragments; 128 -> 64
// (layer 3) 72 fragments per filter tile, so two waves share a (pixel tile, filter tile) by CLASS -- role 0 owns classes {11, 00} (40 fragments, 5 MFMAs per
// 16-step), role 1 {01, 10} (32 fragments, 4 MFMAs): no sums to exchange.  Every class is rounded and stored straight from the accumulators at its output
// parity (output pixel (2 i + ph, 2 j + pw)).
// class c = 2 ph + pw; whether it has a tap with input shift (dh, dw); its taps; whether wave role r owns it (one role: every class; two roles: {3, 0} / {1, 2});
// its first filter fragment (in taps) in the role's register bank
constexpr bool cq_uses(int c, int dh, int dw) { return (dh == 0 || (c >> 1) == 1) && (dw == 0 || (c & 1) == 1); }
constexpr int cq_taps(int c) { return ((c >> 1) + 1) * ((c & 1) + 1); }
constexpr bool cq_owns(int roles, int r, int c) { return roles == 1 || (r == 0 ? (c == 0 || c == 3) : (c == 1 || c == 2)); }
constexpr int cq_base(int roles, int r, int c) {
    int b = 0;
    for (int i = 0; i < c; ++i)
        if (cq_owns(roles, r, i)) b += cq_taps(i);
    return b;
}

template <int CIN, int COUT, int MT> struct CqGeom {
    static constexpr int PXB = CIN * 2, PP = PXB + 16, SLOTS = PP / 16;
    static constexpr int SWP = MT * 32, NPX = SWP + 1;                    // du pixels j .. j + 64 of a row
    static constexpr int XPIECES = (NPX * PP + 1023) / 1024, XROWB = XPIECES * 1024;
    static constexpr int NSLOT = 3;                                       // rows i, i + 1 in use + the one being fetched
    static constexpr int KPT = CIN / 16;                                  // 16-wide reduction steps per tap
    static constexpr int ROLES = 9 * KPT > 36 ? 2 : 1;                    // waves that share a (pixel tile, filter tile) by class
    static constexpr int NT = COUT / 32, NW = MT * NT * ROLES;
    static constexpr int MAXF = (ROLES == 1 ? 9 : 5) * KPT;               // filter fragments of the busiest role
    static_assert(MAXF <= 40, "160 filter registers per lane");
    static constexpr int XBASE = 0, LDS = NSLOT * XROWB;
    static constexpr int XPW = (XPIECES + NW - 1) / NW;
};
template <int CIN, int COUT, int MT> constexpr int cq_threads() { return CqGeom<CIN, COUT, MT>::NW * 64; }
// what differs between the classes is the filter bank and the output parity: one ConvArgs (class 3's) + the four banks keep the scalar registers free
struct CqArgs {
    ConvArgs a;
    const void* w[4];
    int kpad[4];
};

template <typename T, int CIN, int COUT, int MT>
__global__ __launch_bounds__((cq_threads<CIN, COUT, MT>()), 2) void conv_strip_quad_kernel(const CqArgs q) {
#if defined(__HIP_DEVICE_COMPILE__)
    typedef CqGeom<CIN, COUT, MT> G;
    constexpr int PP = G::PP, XROWB = G::XROWB, NW = G::NW, NT = G::NT, KPT = G::KPT, XPW = G::XPW, ROLES = G::ROLES;
    typedef typename Mfma<T>::frag frag;
    typedef __attribute__((address_space(3))) void* lds_ptr_t;
    __shared__ __attribute__((aligned(1024))) unsigned char smem[G::LDS];
    const ConvArgs& p = q.a;   // the geometry every class shares: du, the strip walk (cs_*), Ho x Wo = the class image

    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int role = wv % ROLES, pair = wv / ROLES;
    const int mt = pair / NT, nt = pair % NT;
    const int frow = lane & 31, fk = lane >> 5;
    const int t_begin = blockIdx.x * p.cs_per;
    int t_end = t_begin + p.cs_per;
    if (t_end > p.cs_T) t_end = p.cs_T;
    if (t_begin >= t_end) return;
    const auto rsrc_x = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, (int)p.x_bytes, 0x00020000);
    const auto rsrc_y = __builtin_amdgcn_make_buffer_rsrc((void*)p.y, 0, (int)p.y_bytes, 0x00020000);
    constexpr unsigned OOB = 0xffffffffu;

    // the role's filters: (class c, tap, 16-step) -> fragment (cq_base(c) + tap) KPT + sub
    frag fw[G::MAXF];
    auto load_filters = [&](auto ROLE) {
        constexpr int R = decltype(ROLE)::value;
        static_for<4>([&](auto CC) {
            constexpr int c = decltype(CC)::value;
            if constexpr (cq_owns(ROLES, R, c)) {
                constexpr int nf = cq_taps(c) * KPT, f0 = cq_base(ROLES, R, c) * KPT;
                const T* wrow = (const T*)q.w[c] + (size_t)(nt * 32 + frow) * q.kpad[c] + fk * 8;
                static_for<nf>([&](auto J) { fw[f0 + decltype(J)::value] = *(const frag*)(wrow + 16 * decltype(J)::value); });
            }
        });
    };
    if constexpr (ROLES == 1) load_filters(IC<0>{});
    else if (role == 0) load_filters(IC<0>{});
    else load_filters(IC<1>{});

    int xj[XPW];
    int xoffl[XPW];
#pragma unroll
    for (int i = 0; i < XPW; ++i) {
        const int e = (i * NW + wv) * 64 + lane;
        const int j = e / G::SLOTS, slot = e - j * G::SLOTS;
        xj[i] = (slot < G::SLOTS - 1 && j < G::NPX) ? j : -1;
        xoffl[i] = (j * p.xpitch + slot * 8) * 2;                  // relative to du pixel (row, w0)
    }
    const unsigned char* xfrag = smem + G::XBASE + (mt * 32 + frow) * PP + fk * 16;

    int tn = fdiv(t_begin, p.dv_pw_mul, p.dv_pw_sh);          // host: reciprocal of the class image's rows
    int row = t_begin - tn * p.Ho;
    int img = fdiv(tn, p.dv_h1_mul, p.dv_h1_sh);
    int strip = tn - img * p.cs_strips;
    int cur[2] = {0, 0}, nxt[2] = {0, 0};
    int hp = 0;
    int c_img = 0, c_strip = 0, c_row = 0, n_img = 0, n_strip = 0, n_row = 0;
    auto take = [&]() { const int s = hp; hp = hp == 2 ? 0 : hp + 1; return s; };
    auto request_row = [&](int hin, int slot, int w0) {
        const bool rowok = (unsigned)hin < (unsigned)p.H;
        const long long xbase = (((long long)img * p.H + hin) * p.W + w0) * p.xpitch * 2;
#pragma unroll
        for (int i = 0; i < XPW; ++i) {
            if ((i + 1) * NW <= G::XPIECES || i * NW + wv < G::XPIECES) {
                const bool ok = rowok && xj[i] >= 0 && w0 + xj[i] < p.W;
                const unsigned off = ok ? (unsigned)(xbase + xoffl[i]) : OOB;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_x, (lds_ptr_t)(smem + G::XBASE + slot * XROWB + (i * NW + wv) * 1024), 16, off, 0, 0, 0);
            }
        }
    };
    auto issue = [&](bool fresh) {
        const int w0 = strip * G::SWP;
        if (fresh) {
            nxt[0] = take(); nxt[1] = take();
            request_row(row, nxt[0], w0);
            request_row(row + 1, nxt[1], w0);
        } else {
            nxt[0] = nxt[1]; nxt[1] = take();
            request_row(row + 1, nxt[1], w0);
        }
        n_img = img; n_strip = strip; n_row = row;
        if (++row == p.Ho) {
            row = 0;
            if (++strip == p.cs_strips) { strip = 0; ++img; }
        }
    };

    // one du row: every shifted fragment (dh, dw, sub) is read once and multiplied into each class of the role that has a tap with that shift.
    // Class c = (ph, pw): taps (ih, iw), ih < ph + 1, iw < pw + 1, with dh = (ph && ih == 0), dw = (pw && iw == 0); K index of the class = (ih nw + iw) Cin + ci.
    auto compute = [&](auto ROLE) {
        constexpr int R = decltype(ROLE)::value;
        f32x16 acc[4][1][1];   // (the classes of the other role are never touched: dead)
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[c][0][0][e] = 0.0f;
        const unsigned char* xr[2] = {xfrag + cur[0] * XROWB, xfrag + cur[1] * XROWB};
        static_for<4 * KPT>([&](auto V) {
            constexpr int v = decltype(V)::value;
            constexpr int shift = v / KPT, sub = v % KPT, dh = shift >> 1, dw = shift & 1;
            constexpr bool used = (cq_owns(ROLES, R, 0) && cq_uses(0, dh, dw)) || (cq_owns(ROLES, R, 1) && cq_uses(1, dh, dw)) ||
                                  (cq_owns(ROLES, R, 2) && cq_uses(2, dh, dw)) || (cq_owns(ROLES, R, 3) && cq_uses(3, dh, dw));
            if constexpr (used) {
                const frag xf = *(const frag*)(xr[dh] + dw * PP + sub * 32);
                static_for<4>([&](auto CC) {
                    constexpr int c = decltype(CC)::value, ph = c >> 1, pw = c & 1;
                    if constexpr (cq_owns(ROLES, R, c) && cq_uses(c, dh, dw)) {
                        constexpr int ih = ph ? (dh ? 0 : 1) : 0, iw = pw ? (dw ? 0 : 1) : 0;
                        acc[c][0][0] = Mfma<T>::run(fw[(cq_base(ROLES, R, c) + ih * (pw + 1) + iw) * KPT + sub], xf, acc[c][0][0]);
                    }
                });
            }
        });
        // epilogue, straight from the registers (no bias, activation, residual or statistics in a data gradient; one call per 9 MFMAs, so the LDS transpose
        // and the index arithmetic of epilogue_wave would dominate): pairs rounded to T, the halves swapped between the lane halves as in epilogue_wave --
        // lane (pixel j, fk) then holds the 16-byte chunks 2 gp + fk (filters 8 chunk .. + 7) of its pixel and stores them at (2 i + ph, 2 j + pw)
        const int jj = c_strip * G::SWP + mt * 32 + frow;
        const bool pv = jj < p.Wo;
        static_for<4>([&](auto CC) {
            constexpr int c = decltype(CC)::value, ph = c >> 1, pw = c & 1;
            if constexpr (cq_owns(ROLES, R, c)) {
                const long long opix = ((long long)c_img * p.oH + 2 * c_row + ph) * p.oW + 2 * jj + pw;
                const unsigned ybase = pv ? (unsigned)((opix * p.ypitch + nt * 32) * 2) : 0xffffffffu;
#pragma unroll
                for (int gp = 0; gp < 2; ++gp) {
                    u32x4 ov;
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        const auto sw = __builtin_amdgcn_permlane32_swap(pack2<T>(acc[c][0][0][8 * gp + 2 * h], acc[c][0][0][8 * gp + 2 * h + 1]),
                                                                        pack2<T>(acc[c][0][0][8 * gp + 4 + 2 * h], acc[c][0][0][8 * gp + 4 + 2 * h + 1]), false, false);
                        ov[h] = (unsigned)sw[0];
                        ov[2 + h] = (unsigned)sw[1];
                    }
                    __builtin_amdgcn_raw_buffer_store_b128(ov, rsrc_y, pv ? ybase + (unsigned)((gp * 2 + fk) * 16) : 0xffffffffu, 0, 0);
                }
            }
        });
    };

    issue(true);
    for (int t = t_begin; t < t_end; ++t) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        cur[0] = nxt[0]; cur[1] = nxt[1];
        c_img = n_img; c_strip = n_strip; c_row = n_row;
        const bool more = t + 1 < t_end;
        const bool fresh = more && row == 0;
        if (more && !fresh) issue(false);
        if constexpr (ROLES == 1) compute(IC<0>{});
        else if (role == 0) compute(IC<0>{});
        else compute(IC<1>{});
        if (fresh) {
            __builtin_amdgcn_s_barrier();
            issue(true);
        }
    }
#endif
}

// knob "conv_strip" as above.  Eligibility: the four classes share the geometry (even H, W: checked by the caller), no residual, (du channels, dx channels) =
// (64, 32) or (128, 64), enough du rows per block
static bool cq_plan(const ConvArgs* cls, CsPlan& pl) {
    const long long mode = y3_knob(Y3K_CONV_STRIP);
    const ConvArgs& a = cls[3];
    if (mode == 0 || a.res || a.ups || a.bias) return false;
    if (!((a.Cin == 64 && a.Cout == 32) || (a.Cin == 128 && a.Cout == 64))) return false;
    for (int i = 0; i < 4; ++i)
        if (!cls[i].x_bytes || !cls[i].w_bytes || !cls[i].y_bytes || cls[i].Ho != a.Ho || cls[i].Wo != a.Wo || cls[i].H != a.H || cls[i].W != a.W) return false;
    if (a.Ho != a.H || a.Wo != a.W) return false;   // even gradient sizes: every class image is du's size
    pl.mt = 2;
    pl.strips = (a.Wo + 63) / 64;
    const long long T = (long long)a.N * pl.strips * a.Ho;
    if (T < 1 || T > 0x3fffffffLL) return false;
    const int nblk = y3_cu_count() * (a.Cin == 64 ? 4 : 1);   // (CqGeom::LDS = 30 / 54 KiB, 2 / 8 waves per block, two waves per SIMD)
    if (mode == 1 && T < 24LL * nblk) return false;
    pl.per = mode > 2 ? (int)mode : (int)((T + nblk - 1) / nblk);
    if (pl.per < 1) pl.per = 1;
    pl.blocks = (int)((T + pl.per - 1) / pl.per);
    pl.T = (int)T;
    return true;
}

template <typename T> int launch_cq(ConvArgs* cls, hipStream_t st) {
    CsPlan pl;
    if (!cq_plan(cls, pl)) Y3_FAIL("conv strip (stride-2 data gradient): no plan (internal)");
    CqArgs q;
    for (int i = 0; i < 4; ++i) {
        if (cls[i].ooh != (i >> 1) || cls[i].oow != (i & 1) || cls[i].omul != 2) Y3_FAIL("conv strip (stride-2 data gradient): class order (internal)");
        q.w[i] = cls[i].w;
        q.kpad[i] = cls[i].Kpad;
    }
    ConvArgs& a = cls[3];
    a.cs_strips = pl.strips; a.cs_T = pl.T; a.cs_per = pl.per;
    a.n_ct = 1; a.n_pt = pl.blocks;
    set_divisors(a);
    magic_u31(a.Ho, a.dv_pw_mul, a.dv_pw_sh);
    magic_u31(pl.strips, a.dv_h1_mul, a.dv_h1_sh);
    q.a = a;
    g_last_variant = "strip_quad";
    const dim3 grid((unsigned)pl.blocks);
    if (cls[3].Cin == 64) hipLaunchKernelGGL((conv_strip_quad_kernel<T, 64, 32, 2>), grid, dim3(cq_threads<64, 32, 2>()), 0, st, q);
    else hipLaunchKernelGGL((conv_strip_quad_kernel<T, 128, 64, 2>), grid, dim3(cq_threads<128, 64, 2>()), 0, st, q);
    Y3_CHECK_LAUNCH();
    return 0;
}
