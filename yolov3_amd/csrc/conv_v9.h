// conv_v9.h -- included by conv.hip INSIDE its anonymous namespace, after conv_v7.h (shares ConvArgs, Mfma, epilogue_wave, fdiv, ...).
//
// v9: the 3x3 / stride 1 / pad 1 convolutions with Cout % 256 == 0 and Cin % 32 == 0 (reference models/common.py:57-81 Conv inside
// Bottleneck.cv2, models/yolov3.yaml:23-31 and the 3x3 convs of the head; their data gradients run through the same kernel on the flipped
// bank) with ONE wave per SIMD.
//
// Why (profiles/r03_v7_ablation.txt): in v6 / v7 two waves share a SIMD and alternate MEM (requests + fragment reads + address arithmetic)
// and MMA (16 MFMAs) phases.  Removing the MFMAs leaves 0.39 us per K-step, removing the MEM work leaves 0.52 us (the matrix pipe back to
// back, barriers included), both together take 0.76 us: the MEM instructions of one wave cost matrix-pipe time of its partner, whatever the
// priorities.  What counts is the NUMBER of non-MFMA instructions per MFMA.  v7 issues ~4.4 of them per MFMA; this kernel ~1.2:
//   * 4 waves, wave tile 64 filters x MP*32 pixels (MP = 6 / 7 / 8), block tile 256 filters x MP*32 pixels, accumulators in AGPRs.  A
//     wave's filter rows are its own: each wave stages ITS 64 rows x 32 k (4 KiB per K-step, ring of 3 private stages, two K-steps ahead)
//     and nobody else reads them -- no barrier per K-step.  Only the pixel operand is shared: one barrier per 32-channel block.
//   * pixel operand = halo patch in PADDED-IMAGE order with an 80-byte row pitch: position Q(n, h, w) = (n (H + 1) + h + 1) (W + 2) + w + 1,
//     i.e. one zero row between images and one zero column on either side of a row.  The zeros cost nothing: `buffer_load ... lds` takes a
//     per-lane SOURCE address, lanes of pad positions carry an out-of-range offset and the descriptor's bounds check lands zeros.  Tap
//     (dh, dw) of output pixel m then reads patch row r(m) + dh (W + 2) + dw for EVERY pixel -- no edge masks, no per-tap address
//     arithmetic: three base registers per 32-pixel column block (one per dh), dw * 80 and the k-substep as instruction immediates.
//     80 = 64 data bytes + 16 pad: 16 consecutive rows of a ds_read_b128 lane group fall on 16 distinct bank quads (5 r mod 16), the
//     conflict-free property the XOR swizzle of v6 / v7 bought with an address computation per tap.
//   * the K-step of a wave (4 MP MFMAs): MFMAs of k-substep 0 with the fragment reads of substep 1 and the 4 filter requests of K-step
//     s + 2 between them, then MFMAs of substep 1 with the reads of K-step s + 1 / substep 0 and (taps 0..6) the patch requests of the next
//     channel block between them; one counted s_waitcnt vmcnt(4) per K-step, never 0 inside the loop.
//   * pixel tiles need not be 256 wide: the host picks MP and the VALID pixels per tile (v9_plan) so that tiles x filter tiles fill whole
//     rounds of the CUs -- 12800 k pixels (batch 32 / 64 at 640 x 640) become 200-pixel tiles under MP = 7: 256 / 512 / 1024 / 2048 tiles
//     instead of 200 / 400 / 800 (78 % of the last round) for 7 / 8 of the MFMA work per tile.
//
// LDS (all of the CU's 160 KiB, one block per CU): [4 waves x 3 x 4 KiB filter stages][2 x 54 KiB patch buffers][4 x 1 KiB dump slots for
// request slots that have nothing to fetch].  The epilogue re-uses the wave's own filter stages as its transpose slice.

constexpr int V9_STAGE = 4096;
constexpr int V9_FILT_WAVE = 3 * V9_STAGE;
constexpr int V9_PATCH = 4 * V9_FILT_WAVE;
constexpr int V9_PB = 54 * 1024;
constexpr int V9_DUMP = V9_PATCH + 2 * V9_PB;
constexpr int V9_LDS = V9_DUMP + 4 * 1024;
constexpr int V9_PITCH = 80;
static_assert(V9_LDS == 163840, "the whole LDS of a CU");

// `s_waitcnt vmcnt(N)` through the builtin (the waitcnt pass parses it; an asm statement is invisible to it and it would add its own
// conservative waits in front of the next requests -- profiles/r02_conv_v8.md): simm16 = vmcnt[3:0] | expcnt[6:4] | lgkmcnt[11:8] | vmcnt[5:4] << 14
template <int N> Y3_DEV void v9_wait_vm() { __builtin_amdgcn_s_waitcnt((N & 15) | (7 << 4) | (15 << 8) | ((N >> 4) << 14)); }

// ABL (tools/v9_ablate.py, -DY3_ABLATE builds only; 0 in the shipped library): the K loop without one of its parts, garbage results, only the
// launch time means something.  1: no filter requests; 2: no patch requests; 3: no pixel-fragment reads; 4: no filter-fragment reads; 5: no
// fragment reads; 6: no MFMAs; 7: MFMAs only; 8: no epilogue.  (Arms 9-13 of profiles/r03_v9_ablation.txt -- filter fragments by register loads, other
// read / MFMA interleavings -- were measured with the lab hooks up to commit 3ead9b8 and removed: none of them is a candidate without a fragment-ordered bank.)
template <typename T, int MP, int XQ, int ABL = 0>
__global__ __launch_bounds__(256, 1) void conv_igemm_v9_kernel(const ConvArgs p) {
#if defined(__HIP_DEVICE_COMPILE__)
    constexpr int MC = 2;
    constexpr int NXP = 7 * XQ;   // patch request slots per wave and channel block: XQ in each of taps 0..6
    constexpr int NPASS = (MP + 1) / 2;
    typedef typename Mfma<T>::frag frag;
    typedef __attribute__((address_space(3))) void* lds_ptr_t;
    __shared__ __attribute__((aligned(1024))) unsigned char smem[V9_LDS];   // the ONLY LDS object

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int frow = lane & 31, fk = lane >> 5;
    // filter tile slowest: the blocks of one XCD (consecutive ids after the remap) share a filter tile, its rows stay in that L2
    const int lin = xcd_remap(blockIdx.x, gridDim.x);
    const int ct = fdiv(lin, p.dv_ct_mul, p.dv_ct_sh);   // host: the divisor is n_pt here
    const int pt = lin - ct * p.n_pt;
    const int m0 = pt * p.v9_vp;
    const int m1 = min(m0 + p.v9_vp, p.M);
    const int PW = p.W + 2;

    const auto rsrc_x = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, (int)p.x_bytes, 0x00020000);
    const auto rsrc_w = __builtin_amdgcn_make_buffer_rsrc((void*)p.w, 0, (int)p.w_bytes, 0x00020000);
    constexpr unsigned OOB = 0x80000000u;   // stays out of range when a channel-block offset is added

    int n0, h0, w0;
    pix_coords(m0, p, n0, h0, w0);
    const int Qf = (n0 * (p.H + 1) + h0 + 1) * PW + w0 + 1;   // padded position of the tile's first pixel
    const int Q0 = Qf - PW - 1;                                // patch row 0

    // ---- per-lane sources of this wave's patch pieces (piece q = 4 i + wave; 64 lanes x 16 B: 12.8 patch rows of 4 data slots + 1 pad slot)
    unsigned xsrc[NXP];
#pragma unroll
    for (int i = 0; i < NXP; ++i) {
        const int q = i * 4 + wv;
        const int e = q * 64 + lane;
        const int pos = e / 5, slot = e - pos * 5;
        const int Qa = Q0 + pos;
        const int Qc = Qa > 0 ? Qa : 0;
        const int R = fdiv(Qc, p.dv_pw_mul, p.dv_pw_sh);
        const int C = Qc - R * PW;
        const int n = fdiv(R, p.dv_h1_mul, p.dv_h1_sh);
        const int hh = R - n * (p.H + 1);
        const bool ok = (slot < 4) & (q < p.v9_npiece) & (Qa >= 0) & (C >= 1) & (C <= p.W) & (hh >= 1) & (n < p.N);
        xsrc[i] = ok ? (unsigned)((((n * p.H + hh - 1) * p.W + (C - 1)) * p.xpitch + slot * 8) * 2) : OOB;
    }
    // ---- filter pieces: piece j = rows 16 j .. 16 j + 15 of the wave's 64, lane -> (row, physical slot); the XOR slot swizzle sits on the source
    unsigned woff[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int row = j * 16 + (lane >> 2);
        const int lsl = (lane & 3) ^ ((row >> 2) & 3);
        woff[j] = (unsigned)(((long long)(ct * 256 + wv * 64 + row) * p.Kpad + lsl * 8) * 2);
    }
    // ---- fragment addresses.  Filters: row frow (+ 32 a), k-group fk + 2 kk at physical slot (fk + 2 kk) ^ swizzle(row); stage and `a` are immediates.
    const int a_k0 = wv * V9_FILT_WAVE + frow * 64 + ((fk ^ ((frow >> 2) & 3)) << 4);
    const int a_k1 = a_k0 ^ 32;
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
        for (int dh = 0; dh < 3; ++dh) bb[dh][b] = V9_PATCH + (r + dh * PW) * V9_PITCH + fk * 16;
    }

    // the accumulators start at the bias of their filter (lane holds filters 8g + 4fk + q of each 32-filter tile)
    f32x16 acc[MC][MP];
#pragma unroll
    for (int a = 0; a < MC; ++a)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int cb = ct * 256 + (wv * MC + a) * 32 + 8 * g + 4 * fk;
            f32x4 bz = {0.f, 0.f, 0.f, 0.f};
            if (p.bias && cb + 4 <= p.Cout) bz = *(const f32x4*)(p.bias + cb);
#pragma unroll
            for (int b = 0; b < MP; ++b)
#pragma unroll
                for (int q = 0; q < 4; ++q) acc[a][b][4 * g + q] = bz[q];
        }

    auto dma_w = [&](int kbyte, int st) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_w, (lds_ptr_t)(smem + wv * V9_FILT_WAVE + st * V9_STAGE + j * 1024), 16, woff[j], kbyte, 0, 0);
    };
    auto dma_x = [&](int i, int cbyte, int buf, bool live) {   // slot i of the wave; live = false (wave-uniform): nothing to fetch, the piece goes to the dump slot
        const int q = i * 4 + wv;
        const bool go = live && q < p.v9_npiece;
        const int dst = go ? V9_PATCH + buf * V9_PB + q * 1024 : V9_DUMP + wv * 1024;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_x, (lds_ptr_t)(smem + dst), 16, xsrc[i], cbyte, 0, 0);
    };
    auto mma = [&](const frag (&af)[MC], const frag (&bf)[MP]) {
#pragma unroll
        for (int a = 0; a < MC; ++a)
#pragma unroll
            for (int b = 0; b < MP; ++b) acc[a][b] = Mfma<T>::run(af[a], bf[b], acc[a][b]);
    };

    // ---- prologue: the whole patch of channel block 0, filter tiles of K-steps 0 and 1, fragments of (K-step 0, substep 0)
    const int ncb = p.cin_blocks;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the bias loads share the counter
#pragma unroll
    for (int i = 0; i < NXP; ++i) dma_x(i, 0, 0, true);
    dma_w(0, 0);
    dma_w(p.Cin * 2, 1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    frag A0[MC], B0[MP], A1[MC], B1[MP];
#pragma unroll
    for (int a = 0; a < MC; ++a) A0[a] = *(const frag*)(smem + a_k0 + a * 2048);
#pragma unroll
    for (int b = 0; b < MP; ++b) B0[b] = *(const frag*)(smem + bb[0][b]);

    int bufd = V9_PB;   // what moves the pixel bases to the other patch buffer
    for (int cb = 0; cb < ncb; ++cb) {
        const bool more = cb + 1 < ncb;
        static_for<9>([&](auto TAP) {
            constexpr int tap = decltype(TAP)::value;
            constexpr int dh = tap / 3, dw = tap % 3;
            constexpr int ntap = (tap + 1) % 9, ndh = ntap / 3, ndw = ntap % 3;
            constexpr int tap2 = (tap + 2) % 9;
            // ---- phase 1: MFMAs of substep 0 | fragment reads of substep 1, filter tile of K-step s + 2 (stage (s + 2) % 3 = (tap + 2) % 3: 9 % 3 == 0)
            if constexpr (ABL == 4 || ABL == 5 || ABL == 7) {
#pragma unroll
                for (int a = 0; a < MC; ++a) asm volatile("" : "=v"(A1[a]));
            } else {
#pragma unroll
                for (int a = 0; a < MC; ++a) A1[a] = *(const frag*)(smem + a_k1 + a * 2048 + (tap % 3) * V9_STAGE);
            }
            if constexpr (ABL == 3 || ABL == 5 || ABL == 7) {
#pragma unroll
                for (int b = 0; b < MP; ++b) asm volatile("" : "=v"(B1[b]));
            } else {
#pragma unroll
                for (int b = 0; b < MP; ++b) B1[b] = *(const frag*)(smem + bb[dh][b] + dw * V9_PITCH + 32);
            }
            if constexpr (ABL != 1 && ABL != 7) dma_w((tap2 * p.Cin + (cb + (tap + 2 >= 9 ? 1 : 0)) * 32) * 2, tap2 % 3);
            if constexpr (ABL != 6) mma(A0, B0);
            else {
#pragma unroll
                for (int a = 0; a < MC; ++a) asm volatile("" :: "v"(A0[a]));
#pragma unroll
                for (int b = 0; b < MP; ++b) asm volatile("" :: "v"(B0[b]));
            }
            {
#pragma unroll
                for (int i = 0; i < MC + MP; ++i) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);   // one MFMA
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);   // one LDS read
                }
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);   // one request
                }
                if constexpr (MC * MP - (MC + MP) - 4 > 0) __builtin_amdgcn_sched_group_barrier(0x008, MC * MP - (MC + MP) - 4, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
            // ---- phase 2: MFMAs of substep 1 | filter tile of K-step s + 1 has landed, fragment reads of (K-step s + 1, substep 0), patch requests
            v9_wait_vm<4>();   // everything but the 4 requests of phase 1: filter tile s + 1 (requested a K-step ago), patch pieces of earlier taps
            if constexpr (tap == 8) {
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
            if constexpr (ABL == 4 || ABL == 5 || ABL == 7) {
#pragma unroll
                for (int a = 0; a < MC; ++a) asm volatile("" : "=v"(A0[a]));
            } else {
#pragma unroll
                for (int a = 0; a < MC; ++a) A0[a] = *(const frag*)(smem + a_k0 + a * 2048 + (ntap % 3) * V9_STAGE);
            }
            if constexpr (ABL == 3 || ABL == 5 || ABL == 7) {
#pragma unroll
                for (int b = 0; b < MP; ++b) asm volatile("" : "=v"(B0[b]));
            } else {
#pragma unroll
                for (int b = 0; b < MP; ++b) B0[b] = *(const frag*)(smem + bb[ndh][b] + ndw * V9_PITCH);
            }
            if constexpr (tap < 7 && ABL != 2 && ABL != 7) {
#pragma unroll
                for (int x = 0; x < XQ; ++x) dma_x(tap * XQ + x, (cb + 1) * 64, (cb + 1) & 1, more);
            }
            if constexpr (ABL != 6) mma(A1, B1);
            else {
#pragma unroll
                for (int a = 0; a < MC; ++a) asm volatile("" :: "v"(A1[a]));
#pragma unroll
                for (int b = 0; b < MP; ++b) asm volatile("" :: "v"(B1[b]));
            }
            {
#pragma unroll
                for (int i = 0; i < MC + MP; ++i) {
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
                if constexpr (MC * MP - (MC + MP) - (tap < 7 ? XQ : 0) > 0) __builtin_amdgcn_sched_group_barrier(0x008, MC * MP - (MC + MP) - (tap < 7 ? XQ : 0), 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        });
    }

    // ---- epilogue: passes of 64 pixels through the wave's own (now idle) filter stages; requests past the last K-step may still be landing there
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    unsigned char* slice = smem + wv * V9_FILT_WAVE;
    if constexpr (ABL == 8) {   // keep the accumulators alive, store nothing
#pragma unroll
        for (int a = 0; a < MC; ++a)
#pragma unroll
            for (int b = 0; b < MP; ++b) asm volatile("" :: "v"(acc[a][b]));
        return;
    }
#pragma unroll
    for (int hb = 0; hb < NPASS; ++hb) {
        if constexpr (MP % 2 == 1) {
            if (hb == NPASS - 1) {
                f32x16 part[MC][1];
#pragma unroll
                for (int a = 0; a < MC; ++a) part[a][0] = acc[a][MP - 1];
                epilogue_wave<T, MC, 1>(p, part, slice, ct * 256 + wv * MC * 32, m0 + hb * 64, lane, pt * NPASS + hb, m1);
                continue;
            }
        }
        f32x16 part[MC][2];
#pragma unroll
        for (int a = 0; a < MC; ++a) { part[a][0] = acc[a][2 * hb]; part[a][1] = acc[a][2 * hb + 1]; }
        epilogue_wave<T, MC, 2>(p, part, slice, ct * 256 + wv * MC * 32, m0 + hb * 64, lane, pt * NPASS + hb, m1);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();   // the slice is private to the wave: its reads of one pass precede the writes of the next
    }
#endif
}

// The host's choice of (MP, valid pixels per tile): fewest (rounds of the CU array) x (per-tile cost) over the instantiated wave-tile widths;
// the per-tile cost is MP per K-step plus a prologue / epilogue worth ~10 K-steps of a full tile.
struct V9Plan {
    int mp, vp, n_pt, npiece;
};
static bool v9_plan(const ConvArgs& a, V9Plan& out) {
    const int n_ct = a.Cout / 256, cus = v7_cu_count();
    const int nk = 9 * (a.Cin / 32);
    const int PW = a.W + 2;
    double best = -1.0;
    const int force_mp = (int)y3_knob(Y3K_V9_MP), force_vp = (int)y3_knob(Y3K_V9_VP);
    for (int mp = 8; mp >= 6; --mp) {
        if (force_mp >= 6 && force_mp <= 8 && mp != force_mp) continue;
        const int tp = mp * 32;
        for (int r = 1; r <= 4096; ++r) {
            long long n_pt = (long long)r * cus / n_ct;
            if (n_pt < 1) continue;
            if (n_pt * tp < a.M) continue;
            int vp = (int)((a.M + n_pt - 1) / n_pt);
            if (vp < 1) vp = 1;
            if (force_vp > 0) vp = force_vp < tp ? force_vp : tp;
            // the kernel's LDS holds 54 pieces of patch: (vp - 1) pixels + 2 pad columns per row crossing + a zero row per image crossing + the halo
            const int rc = (vp - 1 + a.W - 1) / a.W, ic = (vp - 1 + a.H * a.W - 1) / (a.H * a.W);
            const int npos = (vp - 1) + 2 * rc + PW * ic + 2 * PW + 3;
            const int npiece = (npos * V9_PITCH + 1023) / 1024;
            if (npiece > 54) break;   // more rounds only shrink vp; a smaller mp may still fit
            const double cost = (double)r * (mp * (nk + 10));
            if (best < 0.0 || cost < best) {
                best = cost;
                out.mp = mp; out.vp = vp; out.n_pt = (a.M + vp - 1) / vp; out.npiece = npiece;
            }
            break;
        }
    }
    return best >= 0.0;
}

static bool v9_eligible(const ConvArgs& a) {
    if (y3_knob(Y3K_CONV_V9) == 0 || a.ups) return false;
    if (a.ks != 3 || a.stride != 1 || a.pad != 1 || a.dil_shift != 0 || a.ntaps != 9 || a.omul != 1 || a.ooh != 0 || a.oow != 0) return false;
    if (a.H != a.Ho || a.W != a.Wo || a.oH != a.Ho || a.oW != a.Wo) return false;
    if ((a.Cin % 32) != 0 || (a.Cout % 256) != 0) return false;
    if (!a.x_bytes || !a.w_bytes || !a.y_bytes || (a.res && !a.r_bytes)) return false;
    for (int t = 0; t < 9; ++t)
        if (a.tdh[t] != t / 3 || a.tdw[t] != t % 3) return false;
    if ((long long)(a.N + 1) * (a.H + 1) * (a.W + 2) >= 0x7fffffffLL) return false;
    // K = 1152 (4 channel blocks per tile): the per-tile prologue + epilogue (~17 us: one wave per SIMD overlaps neither with another tile's
    // K loop) outweigh the filled rounds -- 173 vs 158 us on 128 -> 256 @80x80 against the 128x128 tiles at 3-4 blocks per CU
    // (profiles/r03_conv_lab_v9_first.txt)
    if (a.Cin < 256 && y3_knob(Y3K_CONV_V9) != 2) return false;
    // one tile per block: below a round of tiles the K-split of conv_v7.h (small batches) is what fills the chip
    if ((long long)y3_ceil_div(a.M, 256) * (a.Cout / 256) < 64 && y3_knob(Y3K_CONV_V9) != 2) return false;
    V9Plan pl;
    return v9_plan(a, pl);
}

template <typename T> int launch_v9(ConvArgs& a, hipStream_t st) {
    V9Plan pl;
    if (!v9_plan(a, pl)) Y3_FAIL("conv v9: no tile plan (internal)");
    a.n_ct = a.Cout / 256;
    a.n_pt = pl.n_pt;
    a.v9_vp = pl.vp;
    a.v9_npiece = pl.npiece;
    set_divisors(a);
    magic_u31(a.n_pt, a.dv_ct_mul, a.dv_ct_sh);   // this kernel divides the block id by n_pt
    magic_u31(a.W + 2, a.dv_pw_mul, a.dv_pw_sh);
    magic_u31(a.H + 1, a.dv_h1_mul, a.dv_h1_sh);
    a.cin_blocks = a.Cin / 32;
    a.nk = 9 * a.cin_blocks;
    a.stat_wp = (pl.mp + 1) / 2;   // statistics rows per pixel tile: one per 64-pixel epilogue pass
    g_last_variant = pl.mp == 8 ? "v9_mp8" : (pl.mp == 7 ? "v9_mp7" : "v9_mp6");
    if (a.dry) return 0;
    const long long nb = (long long)a.n_ct * a.n_pt;
    if (nb > 0x7fffffffLL) Y3_FAIL("conv grid too large");
    const dim3 grid((unsigned)nb), block(256);
    const bool two = pl.npiece > 28;
#ifdef Y3_ABLATE
    if (const char* e = getenv("Y3_V9_ABL")) {   // lab build only
        const int abl = atoi(e);
        if (pl.mp == 7 && !two && abl >= 1 && abl <= 8) {
            switch (abl) {
                case 1: hipLaunchKernelGGL((conv_igemm_v9_kernel<T, 7, 1, 1>), grid, block, 0, st, a); break;
                case 2: hipLaunchKernelGGL((conv_igemm_v9_kernel<T, 7, 1, 2>), grid, block, 0, st, a); break;
                case 3: hipLaunchKernelGGL((conv_igemm_v9_kernel<T, 7, 1, 3>), grid, block, 0, st, a); break;
                case 4: hipLaunchKernelGGL((conv_igemm_v9_kernel<T, 7, 1, 4>), grid, block, 0, st, a); break;
                case 5: hipLaunchKernelGGL((conv_igemm_v9_kernel<T, 7, 1, 5>), grid, block, 0, st, a); break;
                case 6: hipLaunchKernelGGL((conv_igemm_v9_kernel<T, 7, 1, 6>), grid, block, 0, st, a); break;
                case 7: hipLaunchKernelGGL((conv_igemm_v9_kernel<T, 7, 1, 7>), grid, block, 0, st, a); break;
                default: hipLaunchKernelGGL((conv_igemm_v9_kernel<T, 7, 1, 8>), grid, block, 0, st, a); break;
            }
            Y3_CHECK_LAUNCH();
            return 0;
        }
    }
#endif
#define Y3_V9(MPV) do { if (two) hipLaunchKernelGGL((conv_igemm_v9_kernel<T, MPV, 2>), grid, block, 0, st, a); \
                        else hipLaunchKernelGGL((conv_igemm_v9_kernel<T, MPV, 1>), grid, block, 0, st, a); } while (0)
    if (pl.mp == 8) Y3_V9(8); else if (pl.mp == 7) Y3_V9(7); else Y3_V9(6);
#undef Y3_V9
    Y3_CHECK_LAUNCH();
    return 0;
}
