// wgrad_strip.h -- included by train.hip after its kernels (shares lds_read_tr16, f32x16, y3_divisor, wgrad_mode, ...); everything here has internal linkage.
//
// Filter gradient of the 3x3 convolutions with few channels on large maps (reference models/yolov3.yaml:15-22, layers 1-4: 32 -> 64 and 64 -> 128 filters at
// 320x320 / 160x160, stride 1 and 2; autograd of models/common.py:75):  dW[co][(tap, ci)] = sum over pixels of du[m][co] * x[m @ tap][ci].
//
// Why another kernel (profiles/r03_wgrad_ring_ab.txt, tools/wgrad_lab.py): the 128 x 128-column tiles of wgrad_dma_kernel stage the x operand once per TAP
// (9 x the bytes of x for a stride-1 layer) and du once per column tile (3-5 x): 960 B (32 -> 64) / 2432 B (64 -> 128) of `buffer_load ... lds` per pixel
// against 192-384 B that exist.  Those launches all sit at 9-10 TB/s of staged bytes -- the L2 -> LDS path, not HBM and not the matrix pipe, is what bounds
// them (and keeping more K-steps in flight changes nothing).  Here every operand byte is staged ONCE:
//   * a block owns ALL 9 Cin columns of dW for 64 filters: 64 x 288 / 64 x 576 fp32 accumulators over 3 / 9 waves (wave = kernel row kh with its three
//     kw tiles of 32 channels, or one tap with its two 32-channel halves; 64 / 96 accumulator registers per lane -- nine waves are three per SIMD, 170
//     registers each).  A 128-filter layer runs two such blocks per strip (blockIdx.y = filter half): x is staged twice, still 2-4 x less than before;
//   * a K-step is one output row of a 64-pixel-wide column strip: 64 rows of du, and the input rows it touches live in a ring of six row buffers --
//     walking down the strip a stride-1 step brings ONE new input row (66 pixels), a stride-2 step two (129 pixels); the nine taps are nine views of the
//     three resident rows (row buffer = kh, pixel shift = kw: instruction immediates).  Padding costs nothing: out-of-image rows / columns are lanes with an
//     out-of-range source offset and the descriptor's bounds check lands zeros;
//   * the MFMA wants the reduction index (pixels) contiguous per lane: ds_read_b64_tr_b16 on the pixel-major rows, as in wgrad_dma_kernel.  Bank conflicts
//     of the four pixel rows of a read group are removed by an XOR on the 16-byte slot inside each 256-byte LDS row, applied on the source side of the DMA:
//     slot ^= 4 (R & mask), mask = 0 / 1 / 1 / 3 for pixel pitch x stride = 64 / 128 / 128 / 256 bytes; R & mask of a lane's reads does not depend on the
//     k-substep (the immediates are multiples of (mask + 1) rows), only on the lane and kw: one base address per (lane, kw);
//   * persistent blocks over the linear K-step index t = ((image, strip), row): one fp32 partial tile per block, summed in block order (deterministic).
// Staged per 64 pixels: 12.4 KiB (32 -> 64, stride 1) instead of 60 KiB.

namespace {

struct StripArgs {
    const void* x;
    const void* du;
    float* part;           // [filter half][block][9 Cin columns][64] fp32
    int N, H, W, xpitch, Ho, Wo, dpitch;
    unsigned x_bytes, du_bytes;
    int strips;            // 64-pixel column strips of an output row
    int T;                 // K-steps in all: N * strips * Ho
    int per;               // K-steps per block
    y3_divisor dv_ho, dv_strips;
};

template <int CIN, int S> struct StripGeom {
    static constexpr int COUT = 64;                           // filters per block
    static constexpr int PXB = CIN * 2;                       // bytes of a pixel of x
    static constexpr int NPX = 63 * S + 3;                    // input pixels under 64 output pixels
    static constexpr int PPP = 1024 / PXB;                    // pixels per 1 KiB request
    static constexpr int NPXA = (NPX + PPP - 1) / PPP * PPP;
    static constexpr int XROWB = NPXA * PXB, XPIECES = XROWB / 1024;
    static constexpr int NSLOT = 6;                           // 3 rows in use + up to 3 being fetched (a fresh strip)
    static constexpr int DROWB = COUT * 2, DTILE = 64 * DROWB, DPIECES = DTILE / 1024;
    static constexpr int TAPW = CIN == 32 ? 3 : 1;            // taps per wave (along kw)
    static constexpr int NW = 9 / TAPW;                       // waves
    static constexpr int NB = TAPW * (CIN / 32), MA = COUT / 32;
    static constexpr int MASKX = PXB == 64 ? (S == 1 ? 0 : 1) : (S == 1 ? 1 : 3);
    static constexpr int XBASE = 2 * DTILE;
    static constexpr int LDS = XBASE + NSLOT * XROWB;
    static constexpr int NCOL = 9 * CIN;
    static_assert(CIN == 32 || CIN == 64, "instantiated shapes");
};

template <int CIN> constexpr int strip_threads() { return (CIN == 32 ? 3 : 9) * 64; }   // (no bare comma inside __launch_bounds__'s macro arguments)
// waves per SIMD the register budget must allow: what the LDS lets a CU hold -- 3 / 2 / 2 / 1 blocks of 3 / 3 / 9 / 9 waves over 4 SIMDs
template <int CIN, int S> constexpr int strip_waves_per_simd() { return CIN == 32 ? (S == 1 ? 3 : 2) : (S == 1 ? 5 : 3); }

template <typename T, int CIN, int S>
__global__ __launch_bounds__(strip_threads<CIN>(), (strip_waves_per_simd<CIN, S>())) void wgrad_strip_kernel(const StripArgs p) {
#if defined(__HIP_DEVICE_COMPILE__)
    typedef StripGeom<CIN, S> G;
    constexpr int COUT = G::COUT;
    constexpr int PXB = G::PXB, XROWB = G::XROWB, DROWB = G::DROWB, DTILE = G::DTILE, NW = G::NW, NB = G::NB, MA = G::MA, TAPW = G::TAPW;
    constexpr int DPW = (G::DPIECES + NW - 1) / NW, XPW = (G::XPIECES + NW - 1) / NW;
    typedef typename std::conditional<std::is_same<T, f16_t>::value, f16x8, bf16x8>::type frag;
    typedef __attribute__((address_space(3))) void* lds_ptr_t;
    __shared__ __attribute__((aligned(1024))) unsigned char smem[G::LDS];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int kh = wv / (3 / TAPW), kw0 = (wv % (3 / TAPW)) * TAPW;
    const int t_begin = blockIdx.x * p.per;
    int t_end = t_begin + p.per;
    if (t_end > p.T) t_end = p.T;
    if (t_begin >= t_end) return;
    const auto rs_x = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, (int)p.x_bytes, 0x00020000);
    const auto rs_d = __builtin_amdgcn_make_buffer_rsrc((void*)p.du, 0, (int)p.du_bytes, 0x00020000);
    constexpr unsigned OOB = 0xffffffffu;

    // ---- staging roles: request q = i NW + wave of an operand, lane -> 16 bytes at LDS position q 1024 + 16 lane; the slot swizzle sits on the source
    int dpx[DPW];
    unsigned doff[DPW];
#pragma unroll
    for (int i = 0; i < DPW; ++i) {
        const int L = (i * NW + wv) * 1024 + lane * 16;
        const int R = L >> 8, ps = (L >> 4) & 15;
        const int ls = ps ^ (4 * (R & 1));           // a 256-byte LDS row = 2 pixels of 64 filters
        const int px = R * 2 + (ls >> 3), chunk = ls & 7;
        dpx[i] = px;
        doff[i] = (unsigned)((px * p.dpitch + (int)blockIdx.y * 64 + chunk * 8) * 2);
    }
    int xj[XPW];
    int xoffl[XPW];
#pragma unroll
    for (int i = 0; i < XPW; ++i) {
        const int L = (i * NW + wv) * 1024 + lane * 16;
        const int R = L >> 8, ps = (L >> 4) & 15;
        const int lin = R * 256 + ((ps ^ (4 * (R & G::MASKX))) << 4);
        const int j = lin / PXB, chunk = (lin % PXB) >> 4;
        xj[i] = j;
        xoffl[i] = ((j - 1) * p.xpitch + chunk * 8) * 2;   // relative to input pixel (row, w0 S)
    }

    // ---- fragment roles (see wgrad_dma_kernel): 16-lane group g reads [4 pixels][16 channels]; lane i: pixel row i >> 2, channels 4 (i & 3) .. + 3
    const int gi = lane & 15, gg = lane >> 4;
    const int krow0 = (gg >> 1) * 8 + (gi >> 2);            // + 16 kk (+ 4 for the second half of a fragment)
    const int chan0 = (gg & 1) * 16 + 4 * (gi & 3);
    const unsigned lds0 = (unsigned)(uintptr_t)(lds_ptr_t)smem;
    unsigned fa[MA], fb[NB];
#pragma unroll
    for (int a = 0; a < MA; ++a) {
        const int cha = a * 32 + chan0;
        const int swz = 4 * ((krow0 >> 1) & 1);
        fa[a] = lds0 + krow0 * DROWB + (((cha >> 3) ^ swz) << 4) + (cha & 4) * 2;
    }
#pragma unroll
    for (int b = 0; b < NB; ++b) {
        const int kw = kw0 + (TAPW == 3 ? b : 0), cib = TAPW == 3 ? 0 : b;
        const int chb = cib * 32 + chan0;
        const int lin0 = (S * krow0 + kw) * PXB + ((chb >> 3) << 4);
        const int R0 = lin0 >> 8, s0 = (lin0 >> 4) & 15;
        fb[b] = lds0 + G::XBASE + R0 * 256 + ((s0 ^ (4 * (R0 & G::MASKX))) << 4) + (chb & 4) * 2;
    }

    f32x16 acc[MA][NB];
#pragma unroll
    for (int a = 0; a < MA; ++a)
#pragma unroll
        for (int b = 0; b < NB; ++b)
#pragma unroll
            for (int q = 0; q < 16; ++q) acc[a][b][q] = 0.0f;

    auto tr_frag = [&](unsigned addr, auto off_c, auto step_c) -> frag {
        constexpr int OFF = decltype(off_c)::value, STEP = decltype(step_c)::value;
        const s16x4_t v0 = lds_read_tr16<OFF>(addr), v1 = lds_read_tr16<OFF + STEP>(addr);
        const s16x8_t r = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
        return __builtin_bit_cast(frag, r);
    };
    auto mma = [&](const frag (&af)[MA], const frag (&bf)[NB]) {
#pragma unroll
        for (int a = 0; a < MA; ++a)
#pragma unroll
            for (int b = 0; b < NB; ++b) {
                if constexpr (std::is_same<T, f16_t>::value) acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[a], bf[b], acc[a][b], 0, 0, 0);
                else acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[a], bf[b], acc[a][b], 0, 0, 0);
            }
    };
    auto landed = [&](frag (&af)[MA], frag (&bf)[NB]) {   // the fragments as operands of the wait: nothing consumes them earlier
        if constexpr (NB == 3) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(af[0]), "+v"(af[1]), "+v"(bf[0]), "+v"(bf[1]), "+v"(bf[2]) : : "memory");
        else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(af[0]), "+v"(af[1]), "+v"(bf[0]), "+v"(bf[1]) : : "memory");
    };
    // one K-step: 64 pixels = 4 k-substeps; xrow = byte offset of the wave's input row buffer
    auto compute = [&](auto stage_c, unsigned xrow) {
        constexpr int ST = decltype(stage_c)::value;
        frag a0[MA], b0[NB], a1[MA], b1[NB];
        unsigned xb[NB];
#pragma unroll
        for (int b = 0; b < NB; ++b) xb[b] = fb[b] + xrow;
        auto rd = [&](auto kk_c, frag (&af)[MA], frag (&bf)[NB]) {
            constexpr int KK = decltype(kk_c)::value;
#pragma unroll
            for (int a = 0; a < MA; ++a) af[a] = tr_frag(fa[a], std::integral_constant<int, ST * DTILE + KK * 16 * DROWB>{}, std::integral_constant<int, 4 * DROWB>{});
#pragma unroll
            for (int b = 0; b < NB; ++b) bf[b] = tr_frag(xb[b], std::integral_constant<int, KK * 16 * S * PXB>{}, std::integral_constant<int, 4 * S * PXB>{});
        };
        rd(std::integral_constant<int, 0>{}, a0, b0);
        landed(a0, b0);
        rd(std::integral_constant<int, 1>{}, a1, b1);
        mma(a0, b0);
        landed(a1, b1);
        rd(std::integral_constant<int, 2>{}, a0, b0);
        mma(a1, b1);
        landed(a0, b0);
        rd(std::integral_constant<int, 3>{}, a1, b1);
        mma(a0, b0);
        landed(a1, b1);
        mma(a1, b1);
    };

    // ---- the walk: (image, strip, row) of the K-step that is requested next; ring slots of the three input rows of the current / next K-step
    int tn = y3_fdiv(t_begin, p.dv_ho);             // (image, strip) pair index
    int row = t_begin - tn * p.Ho;
    int img = y3_fdiv(tn, p.dv_strips);
    int strip = tn - img * p.strips;
    int cur[3] = {0, 0, 0}, nxt[3] = {0, 0, 0};
    int hp = 0;                                      // next ring slot to hand out
    auto take = [&]() { const int s = hp; hp = hp + 1 == G::NSLOT ? 0 : hp + 1; return s; };

    auto issue = [&](int t, int stage) {
        const bool fresh = t == t_begin || row == 0;
        int first_new;                               // input rows kh >= first_new are fetched, the others stay where they are
        if (fresh) { nxt[0] = take(); nxt[1] = take(); nxt[2] = take(); first_new = 0; }
        else if (S == 1) { nxt[0] = nxt[1]; nxt[1] = nxt[2]; nxt[2] = take(); first_new = 2; }
        else { nxt[0] = nxt[2]; nxt[1] = take(); nxt[2] = take(); first_new = 1; }
        const int w0 = strip * 64;
        const unsigned dbase = (unsigned)((((long long)img * p.Ho + row) * p.Wo + w0) * p.dpitch * 2);
#pragma unroll
        for (int i = 0; i < DPW; ++i) {
            if ((i + 1) * NW <= G::DPIECES || i * NW + wv < G::DPIECES) {
                const unsigned off = (w0 + dpx[i] < p.Wo) ? dbase + doff[i] : OOB;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_d, (lds_ptr_t)(smem + stage * DTILE + (i * NW + wv) * 1024), 16, off, 0, 0, 0);
            }
        }
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            if (r < first_new) continue;
            const int hin = row * S - 1 + r;
            const bool rowok = (unsigned)hin < (unsigned)p.H;
            const long long xbase = (((long long)img * p.H + hin) * p.W + w0 * S) * p.xpitch * 2;
#pragma unroll
            for (int i = 0; i < XPW; ++i) {
                if ((i + 1) * NW <= G::XPIECES || i * NW + wv < G::XPIECES) {
                    const int col = w0 * S - 1 + xj[i];
                    const bool ok = rowok && xj[i] < G::NPX && (unsigned)col < (unsigned)p.W;
                    const unsigned off = ok ? (unsigned)(xbase + xoffl[i]) : OOB;
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_x, (lds_ptr_t)(smem + G::XBASE + nxt[r] * XROWB + (i * NW + wv) * 1024), 16, off, 0, 0, 0);
                }
            }
        }
        // advance to the K-step after t
        if (++row == p.Ho) {
            row = 0;
            if (++strip == p.strips) { strip = 0; ++img; }
        }
    };

    issue(t_begin, 0);
    for (int t = t_begin; t < t_end; t += 2) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();   // K-step t has landed for every wave; nobody reads the buffers of K-step t - 1 any more
        cur[0] = nxt[0]; cur[1] = nxt[1]; cur[2] = nxt[2];
        if (t + 1 < t_end) issue(t + 1, 1);
        compute(std::integral_constant<int, 0>{}, (unsigned)(cur[kh] * XROWB));
        if (t + 1 >= t_end) break;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        cur[0] = nxt[0]; cur[1] = nxt[1]; cur[2] = nxt[2];
        if (t + 2 < t_end) issue(t + 2, 0);
        compute(std::integral_constant<int, 1>{}, (unsigned)(cur[kh] * XROWB));
    }

    // ---- D[row = co][col = (tap, ci)] -> partial tile [column][co] (each lane owns 4 consecutive co: one 16-byte store)
    const int frow = lane & 31, fk = lane >> 5;
    float* tile = p.part + ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * (G::NCOL * COUT);
#pragma unroll
    for (int b = 0; b < NB; ++b) {
        const int kw = kw0 + (TAPW == 3 ? b : 0), cib = TAPW == 3 ? 0 : b;
        const int nl = (kh * 3 + kw) * CIN + cib * 32 + frow;
#pragma unroll
        for (int a = 0; a < MA; ++a)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int co = a * 32 + 8 * g + 4 * fk;
                f32x4 v = {acc[a][b][4 * g], acc[a][b][4 * g + 1], acc[a][b][4 * g + 2], acc[a][b][4 * g + 3]};
                *(f32x4*)(tile + nl * COUT + co) = v;
            }
    }
#endif
}

// dW (OIHW fp32) = sum over the blocks' partial tiles; thread = (filter half, column, 4 filters) x one of G groups of consecutive blocks (train.hip::wgrad_reduce4_kernel:
// the partial tiles of 768 blocks summed by 4608 threads in 18 blocks took 0.26-0.65 ms per launch, 1.3 ms per step); group sums added in group order: fixed order
__global__ __launch_bounds__(256) void wgrad_strip_reduce_kernel(const float* __restrict__ part, int blocks, int cin, int cout, float* __restrict__ dw, int G) {
    __shared__ f32x4 red[256];
    const int U = 256 / G, u = (int)threadIdx.x % U, g = (int)threadIdx.x / U;
    const int ncol = 9 * cin;
    const int idx = blockIdx.x * U + u;
    const bool live = idx < ncol * (cout / 4);
    const int half = idx / (ncol * 16), rem = idx - half * (ncol * 16);
    const int col = rem >> 4, co = (rem & 15) * 4;
    const size_t stride = (size_t)ncol * 64;
    f32x4 a = {0.f, 0.f, 0.f, 0.f};
    if (live) {
        const float* src = part + (size_t)half * blocks * stride + (size_t)col * 64 + co;
        const int per = (blocks + G - 1) / G;
        int b = g * per;
        const int b_end = b + per < blocks ? b + per : blocks;
        for (; b + 8 <= b_end; b += 8) {
            f32x4 v[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) v[q] = __builtin_nontemporal_load((const f32x4*)(src + (size_t)(b + q) * stride));
#pragma unroll
            for (int q = 0; q < 8; ++q) a += v[q];
        }
        for (; b < b_end; ++b) a += __builtin_nontemporal_load((const f32x4*)(src + (size_t)b * stride));
    }
    if (G > 1) {
        red[threadIdx.x] = a;
        __syncthreads();
        if (g != 0) return;
        for (int q = 1; q < G; ++q) a += red[q * U + u];
    }
    if (!live) return;
    const int tap = col / cin, ci = col - tap * cin;
#pragma unroll
    for (int q = 0; q < 4; ++q) dw[((size_t)(half * 64 + co + q) * cin + ci) * 9 + tap] = a[q];
}

}  // namespace

struct StripPlan {
    int blocks, per, strips, T;
    size_t ws_bytes;
};
// the layers this kernel serves: 3x3, pad 1, stride 1 / 2, (32 -> 64) or (64 -> 128) unpadded channels, half precision, no bias gradient, enough K-steps
// for every block to amortise its partial tile (knob "wgrad_strip": 1 on, 0 off, 2 also small launches, N > 2 also small launches with N K-steps per block -- tests)
static bool strip_plan(const y3_conv_desc* d, int n, int h, int w, int cout_real, int cin_real, bool want_dbias, StripPlan& pl) {
    const long long mode = y3_knob(Y3K_WGRAD_STRIP);
    if (mode == 0 || wgrad_mode() != 0 || want_dbias) return false;
    if (d->dtype != Y3_F16 && d->dtype != Y3_BF16) return false;
    if (d->ksize != 3 || (d->stride != 1 && d->stride != 2)) return false;
    if (!((d->cin == 32 && d->cout == 64) || (d->cin == 64 && d->cout == 128)) || cin_real != d->cin || cout_real != d->cout) return false;
    const int Ho = (h + 2 - 3) / d->stride + 1, Wo = (w + 2 - 3) / d->stride + 1;
    const int strips = (Wo + 63) / 64;
    const long long T = (long long)n * strips * Ho;
    if (T > 0x3fffffffLL || T < 1) return false;
    const int halves = d->cout / 64;
    const int per_cu = d->cin == 32 ? (d->stride == 1 ? 3 : 2) : (d->stride == 1 ? 2 : 1);   // blocks the LDS of a CU holds (StripGeom::LDS)
    int nblk = y3_cu_count() * per_cu / halves;   // per filter half
    if (nblk < 64) nblk = 64;
    if (mode == 1 && T < 16LL * nblk) return false;   // at least 16 K-steps per block
    pl.per = mode > 2 ? (int)mode : (int)((T + nblk - 1) / nblk);   // (knob > 2: that many K-steps per block -- tests walk strip / image boundaries inside a block)
    if (pl.per < 1) pl.per = 1;
    pl.blocks = (int)((T + pl.per - 1) / pl.per);
    pl.strips = strips;
    pl.T = (int)T;
    pl.ws_bytes = (size_t)pl.blocks * halves * 9 * d->cin * 64 * sizeof(float);
    return true;
}

template <typename T> static void launch_strip_t(const y3_conv_desc* d, const StripArgs& a, int blocks, hipStream_t st) {
    const dim3 grid((unsigned)blocks, (unsigned)(d->cout / 64));
    if (d->cin == 32) {
        if (d->stride == 1) hipLaunchKernelGGL((wgrad_strip_kernel<T, 32, 1>), grid, dim3(StripGeom<32, 1>::NW * 64), 0, st, a);
        else hipLaunchKernelGGL((wgrad_strip_kernel<T, 32, 2>), grid, dim3(StripGeom<32, 2>::NW * 64), 0, st, a);
    } else {
        if (d->stride == 1) hipLaunchKernelGGL((wgrad_strip_kernel<T, 64, 1>), grid, dim3(StripGeom<64, 1>::NW * 64), 0, st, a);
        else hipLaunchKernelGGL((wgrad_strip_kernel<T, 64, 2>), grid, dim3(StripGeom<64, 2>::NW * 64), 0, st, a);
    }
}
