// wgrad_patch.h -- included by train.hip after its kernels (shares lds_read_tr16, f32x16, y3_divisor, ...); everything here has internal linkage.
//
// Filter gradient of the 3x3 / stride 1 / pad 1 convolutions with Cin % 64 == 0 and Cout % 128 == 0 (reference models/common.py:75 Conv inside
// Bottleneck.cv2 -- models/yolov3.yaml:23-31 -- and the 3x3 convs of the head; autograd of conv2d behind train.py:411):
//     dW[co][ci][kh][kw] = sum over pixels m of du[m][co] * x[m + (kh - 1, kw - 1)][ci].
// It replaces wgrad_big_kernel (256 x 256 tiles over an im2col'd column axis) on those layers.  That kernel staged x once per TAP COLUMN TILE (the nine (tap,
// channel) column tiles of a filter tile read x rows W pixels apart: 479 MB fetched per launch for 160-260 MB of operands), issued 48 transposing LDS reads per 16
// MFMAs from 8 waves (the LDS array as busy as the matrix pipe) and had two block barriers per 32 pixels.  Here the x operand follows conv_v10.h's scheme:
//   * THE REDUCTION AXIS IS THE PADDED POSITION q = (n (H + 1) + h + 1)(W + 2) + w + 1 (one zero row between images, one zero column either side of a row).  Both
//     operands are staged in that order -- du_p[q] and x_p[q] are zero at pad positions, landed by out-of-range `buffer_load ... lds` lanes -- so tap (dh, dw) of EVERY
//     position is x_p[q + dh (W + 2) + dw]: a constant row offset.  No edge masks, no per-tap staging: x is staged ONCE per (block, position), the nine taps are nine
//     views of it.  Price: the pad positions are multiplied too (3.8 % of the MFMAs at 80 x 80, 7.6 % at 40 x 40, 15.5 % at 20 x 20 -- zeros, which cost issue slots but
//     little power).
//   * x lives in a RING of 512 positions per 32-channel plane, 64-byte rows (a transposing read group covers 4 rows x 64 B = 256 contiguous bytes whatever the row offset:
//     conflict-free for every tap without a swizzle).  A wave keeps one read address per (tap row dh, k-group) -- twelve per stage, each reduced mod 512 rows when the
//     stage's base moves (two VALU operations, in MFMA gaps that carry no reads) -- and reaches (dw, k-half) by instruction immediates of at most 6 rows; the ring's first
//     16 rows are mirrored behind it for those (one stage in eight issues its x requests twice).  (First cut: three addresses per stage and a mirror of 64 + 2 BACK rows --
//     at 80 x 80 every second x request was issued twice and the others went to a dump slot to keep the request count fixed: 8 requests per wave and stage instead of 6.)
//   * block tile = 128 filters x (9 taps x 64 channels); 4 waves, ONE PER SIMD, wave tile 64 filters x 9 taps x 32 channels = 288 accumulator registers: per
//     16 positions a wave reads 2 + 9 fragments (22 ds_read_b64_tr_b16) for 18 MFMAs (wgrad_big: 24 for 16, from twice the waves), and a block stages 24 KiB per
//     64 positions (288 MFMAs) where wgrad_big staged 32 KiB per 32 pixels (128 MFMAs): a third of the bytes per MFMA.
//   * ONE block barrier per 64 positions (72 MFMAs per wave), placed between k-groups 2 and 3 of a stage: every fragment of the stage is in registers by then, so the
//     stage's du buffer is free for the requests of stage s + 3 (three stages in flight) and the first fragments of stage s + 1 are read under k-group 3's MFMAs --
//     no exposed read after the barrier.  Counted `vmcnt`: a wave issues 6 requests per stage (4 du planes, 2 x planes; 8 when its x group is the mirrored one) and waits
//     for all but the youngest stage's.
//   * split-K over contiguous position ranges, one fp32 slab per block in REGISTER order (every store instruction writes 1 KiB contiguous), summed in slice order
//     by wgrad_patch_reduce_kernel: deterministic, no atomics.  The blocks of a slice are neighbours on one XCD (xcd_remap) and walk the same positions: the operand
//     rows they share meet in that L2.
//   * MEASURED AND NOT KEPT (round 6): reading the x positions of a tap row once (three groups of four positions) and cutting the three dw taps out of them in registers
//     (four v_perm_b32 for dw = 0, three v_mov_b32 for dw = +1): 13 transposing reads per k-group instead of 22 -- and 3 % SLOWER per launch on the same box (247 / 244 / 252
//     -> 254 / 252 / 261 us, profiles/r06_wgrad_patch_lab.txt): the VALU results feed MFMAs directly (two wait states each) and what this loop pays for is every
//     instruction that sits between two MFMAs, not the LDS array's cycles.
// LDS (129 KiB, one block per CU; the rings start at multiples of 32 KiB so that `& 0x7fc0` is the wrap): [plane 0: 32 KiB ring + 1 KiB mirror][du stage 0]
// [plane 1 at 64 KiB: ring + mirror][du stages 1, 2].

namespace {

struct PatchArgs {
    const void* x;
    const void* du;
    float* part;            // [slice][tile][wave][a][tap][g][lane][4] fp32
    int N, H, W, xpitch, dpitch;
    unsigned x_bytes, du_bytes;
    int PW, PH;             // W + 2, H + 1
    int back16;             // ceil16(W + 3): positions the x ring keeps behind (and requests ahead of) the du stage
    int Qc;                 // padded positions walked in all (multiple of 64)
    int per;                // positions per slice (multiple of 64)
    int tiles, n_cit;       // (Cout / 128) (Cin / 64) block tiles, Cin / 64 of them per filter tile
    y3_divisor dv_pw, dv_ph, dv_tiles, dv_cit;
};

constexpr int WP_KS = 64;                       // positions per stage
constexpr int WP_NST = 3;                       // du stages
constexpr int WP_DU_STAGE = WP_KS * 128 * 2;    // 16 KiB: 4 planes of [64 rows][32 filters]
constexpr int WP_RING = 512;
constexpr int WP_MAX_BACK = 144;                // 3 stages of requests + the window of a stage (64 + 2 back) fit the ring
constexpr int WP_PLANE = 65536;                 // plane 1 - plane 0
constexpr int WP_MIRROR = WP_RING * 64;         // the ring's first 16 rows again, behind it
constexpr int WP_DU0 = WP_MIRROR + 2048, WP_DU1 = WP_PLANE + WP_MIRROR + 1024, WP_DU2 = WP_DU1 + WP_DU_STAGE;
constexpr int WP_LDS = WP_DU2 + WP_DU_STAGE;
constexpr int WP_SLAB = 128 * 576;              // floats per block
static_assert(WP_LDS <= 163840 && WP_DU0 + WP_DU_STAGE <= WP_PLANE, "the LDS of a CU");

template <int I> struct WIC {
    static constexpr int value = I;
};
template <typename F, int... Is> Y3_DEV void wp_for_impl(F&& f, std::integer_sequence<int, Is...>) { (f(WIC<Is>{}), ...); }
template <int N, typename F> Y3_DEV void wp_for(F&& f) { wp_for_impl(f, std::make_integer_sequence<int, N>{}); }
template <int N> Y3_DEV void wp_wait_vm() { __builtin_amdgcn_s_waitcnt((N & 15) | (7 << 4) | (15 << 8) | ((N >> 4) << 14)); }

// ABL (tools/wgrad_patch_ablate.py, -DY3_ABLATE builds only; 0 in the shipped library): the kernel without one of its parts -- garbage results, only the launch time
// means something.  1: no requests in the loop; 2: no fragment reads in the loop; 3: no barrier / counted wait; 4: no request sources, no requests; 5: MFMAs only (2 + 3 + 4);
// 6: no slab stores; 7: no MFMAs
template <typename T, int ABL = 0>
__global__ __launch_bounds__(256, 1) void wgrad_patch_kernel(const PatchArgs p) {
#if defined(__HIP_DEVICE_COMPILE__)
    typedef typename std::conditional<std::is_same<T, f16_t>::value, f16x8, bf16x8>::type frag;
    typedef __attribute__((address_space(3))) void* lds_ptr_t;
    __shared__ __attribute__((aligned(1024))) unsigned char smem[WP_LDS];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wc = wv >> 1, cib = wv & 1;       // wave = filters 64 wc .. + 63 x channels 32 cib .. + 31 of the block tile, all nine taps
    const int L = xcd_remap(blockIdx.x, gridDim.x);
    const int slice = y3_fdiv(L, p.dv_tiles), tile = L - slice * p.tiles;
    const int cot = y3_fdiv(tile, p.dv_cit), cit = tile - cot * p.n_cit;
    const int q0 = slice * p.per;
    const int q1 = min(q0 + p.per, p.Qc);
    const int S = (q1 - q0) / WP_KS;            // host: every slice has at least one stage
    const auto rs_x = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, (int)p.x_bytes, 0x00020000);
    const auto rs_d = __builtin_amdgcn_make_buffer_rsrc((void*)p.du, 0, (int)p.du_bytes, 0x00020000);
    constexpr unsigned OOB = 0x80000000u;       // stays out of range when a plane offset is added

    // ---- requests.  Lane = 4 r + c: row r of a 16-row group, 16-byte chunk c (8 channels) of the plane's 64-byte row.
    const int rrow = lane >> 2, rchunk = lane & 3;
    const int g0 = p.back16 >> 3;               // x groups (of 16 positions) between a stage's first group and the group its requests fetch
    auto src_of = [&](int q, bool extra_ok, int pitch, int chan) -> unsigned {   // byte offset of (padded position q, channel chan), OOB for pad positions
        const int qc = q > 0 ? q : 0;
        const int R = (int)(__umulhi((unsigned)qc, p.dv_pw.mul) >> (p.dv_pw.sh - 1)), wp = qc - R * p.PW;   // (W + 2 and H + 1 are never 1: no mul == 0 form, no branch)
        const int n = (int)(__umulhi((unsigned)R, p.dv_ph.mul) >> (p.dv_ph.sh - 1)), hp = R - n * p.PH;
        const bool ok = extra_ok & (q >= 0) & (wp >= 1) & (wp <= p.W) & (hp >= 1) & (n < p.N);
        unsigned off = (unsigned)((((n * p.H + hp - 1) * p.W + wp - 1) * pitch + chan) * 2);
        asm volatile("" : "+v"(off));   // (computed for every lane: a select, not a branch around three multiplications)
        return ok ? off : OOB;
    };
    // the requests of a stage, as pieces that ride in MFMA gaps: sources first (VALU), then six or eight requests
    unsigned rq_du = OOB, rq_x = OOB;
    unsigned char *rq_dd = smem, *rq_d0 = smem;
    bool rq_mir = false;                        // wave-uniform: this x group is the ring's first one, whose rows exist twice
    auto du_stage_off = [&](int s) -> int {
        const int r = s % WP_NST;
        return r == 0 ? WP_DU0 : (r == 1 ? WP_DU1 : WP_DU2);
    };
    auto rq_prep_du = [&](int s) {              // this wave's 16 positions of stage s
        const int q = q0 + s * WP_KS + 16 * wv + rrow;
        rq_du = src_of(q, q < q1, p.dpitch, cot * 128 + rchunk * 8);
        rq_dd = smem + du_stage_off(s) + wv * 1024;
    };
    auto rq_prep_x = [&](int g) {               // x positions q0 - back16 + 16 g .. + 15 into ring rows 16 (g mod 32)
        rq_x = src_of(q0 - p.back16 + 16 * g + rrow, true, p.xpitch, cit * 64 + rchunk * 8);
        const int gi = g & (WP_RING / 16 - 1);
        rq_mir = gi == 0;
        rq_d0 = smem + gi * 1024;
    };
    auto rq_issue = [&](auto K) {
        constexpr int k = decltype(K)::value;
        if constexpr (k < 4) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_d, (lds_ptr_t)(rq_dd + k * 4096), 16, rq_du, k * 64, 0, 0);   // the four 32-filter planes
        else if constexpr (k == 4) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_x, (lds_ptr_t)rq_d0, 16, rq_x, 0, 0, 0);
        else if constexpr (k == 5) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_x, (lds_ptr_t)(rq_d0 + WP_PLANE), 16, rq_x, 64, 0, 0);
        else if constexpr (k == 6) {
            if (rq_mir) {
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_x, (lds_ptr_t)(rq_d0 + WP_MIRROR), 16, rq_x, 0, 0, 0);
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_x, (lds_ptr_t)(rq_d0 + WP_MIRROR + WP_PLANE), 16, rq_x, 64, 0, 0);
            }
        }
    };
    auto dma_x_group = [&](int g) {
        rq_prep_x(g);
        wp_for<3>([&](auto K) { rq_issue(WIC<4 + decltype(K)::value>{}); });
    };
    auto dma_stage = [&](int s) {
        rq_prep_du(s);
        rq_prep_x(4 * s + g0 + wv);
        wp_for<7>([&](auto K) { rq_issue(K); });
    };

    // ---- fragments (see wgrad_dma_kernel): 16-lane group gg reads channel block 16 (gg & 1) of a 32-wide MFMA tile for k-group gg >> 1; lane gi of the group addresses
    // position row gi >> 2, channels 4 (gi & 3) .. + 3 and receives channel gi's 4 positions; two reads (rows + 0, + 4) make one 8-k fragment
    const int gi = lane & 15, gg = lane >> 4;
    const int krow0 = (gg >> 1) * 8 + (gi >> 2);
    const int chan0 = (gg & 1) * 16 + 4 * (gi & 3);
    const unsigned lds0 = (unsigned)(uintptr_t)(lds_ptr_t)smem;
    const unsigned lane_a = lds0 + (2 * wc) * 4096 + krow0 * 64 + chan0 * 2;                       // + stage, a * 4096, kg * 1024, half * 256
    // x: row (ring base + 16 kg + back16 - 1 + (dh - 1) PW + krow0) mod 512 of the wave's plane; (dw, half) are immediates of 0 .. 6 rows (the mirror's reach)
    unsigned rc6[3];
#pragma unroll
    for (int dh = 0; dh < 3; ++dh) rc6[dh] = (unsigned)((p.back16 - 1 + (dh - 1) * p.PW + krow0) << 6);
    const unsigned lane_x = lds0 + cib * WP_PLANE + chan0 * 2;
    auto b_addr_of = [&](int s, int kg, int dh) -> unsigned {
        const unsigned t6 = (unsigned)((((s * WP_KS) + 16 * kg) & (WP_RING - 1)) << 6);
        return ((rc6[dh] + t6) & 0x7fc0u) + lane_x;
    };

    f32x16 acc[2][9];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int q = 0; q < 16; ++q) acc[a][t][q] = 0.0f;

    struct Frags {
        s16x4_t al[2], ah[2], bl[9], bh[9];
    };
    // read op r of a k-group: 0..3 the du fragments (a = r >> 1, half = r & 1), 4..21 the x fragments (tap = (r - 4) >> 1)
    auto read_op = [&](auto R, auto KG, Frags& f, unsigned a_addr, const unsigned (&b_addr)[3]) {   // b_addr: the three tap-row addresses of THIS k-group
        constexpr int r = decltype(R)::value, kg = decltype(KG)::value;
        if constexpr (r < 4) {
            constexpr int a = r >> 1, half = r & 1;
            const s16x4_t v = lds_read_tr16<a * 4096 + kg * 1024 + half * 256>(a_addr);
            if constexpr (half) f.ah[a] = v; else f.al[a] = v;
        } else {
            constexpr int t = (r - 4) >> 1, half = r & 1, dh = t / 3, dw = t % 3;
            const s16x4_t v = lds_read_tr16<dw * 64 + half * 256>(b_addr[dh]);
            if constexpr (half) f.bh[t] = v; else f.bl[t] = v;
        }
    };
    auto landed = [&](Frags& f) {   // the fragments as operands of the wait: nothing consumes them earlier
        asm volatile("s_waitcnt lgkmcnt(0)"
                     : "+v"(f.al[0]), "+v"(f.ah[0]), "+v"(f.al[1]), "+v"(f.ah[1]), "+v"(f.bl[0]), "+v"(f.bh[0]), "+v"(f.bl[1]), "+v"(f.bh[1]), "+v"(f.bl[2]), "+v"(f.bh[2]),
                       "+v"(f.bl[3]), "+v"(f.bh[3]), "+v"(f.bl[4]), "+v"(f.bh[4])
                     :
                     : "memory");
        asm volatile("" : "+v"(f.bl[5]), "+v"(f.bh[5]), "+v"(f.bl[6]), "+v"(f.bh[6]), "+v"(f.bl[7]), "+v"(f.bh[7]), "+v"(f.bl[8]), "+v"(f.bh[8]) : : "memory");
    };
    auto mk = [&](const s16x4_t lo, const s16x4_t hi) -> frag {
        const s16x8_t r = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        return __builtin_bit_cast(frag, r);
    };
    auto mfma = [&](auto I, const Frags& f) {
        constexpr int i = decltype(I)::value, a = i / 9, t = i % 9;
        const frag af = mk(f.al[a], f.ah[a]), bf = mk(f.bl[t], f.bh[t]);
        if constexpr (ABL == 7) {
            asm volatile("" :: "v"(af), "v"(bf));
        } else if constexpr (i >= 16) {
            // 288 accumulator registers against 256 AGPRs: through the builtin the compiler keeps EVERY accumulator in the AGPR file and rotates the surplus through
            // VGPRs (first build: 832 v_accvgpr_* per 72 MFMAs).  The last two tiles therefore live in VGPRs, multiplied by the VGPR form of the instruction as asm
            // (nothing else touches their registers between two of these; the epilogue reads them behind explicit wait states).
            if constexpr (std::is_same<T, f16_t>::value) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc[a][t]) : "v"(af), "v"(bf));
            else asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc[a][t]) : "v"(af), "v"(bf));
        } else {
            if constexpr (std::is_same<T, f16_t>::value) acc[a][t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af, bf, acc[a][t], 0, 0, 0);
            else acc[a][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, bf, acc[a][t], 0, 0, 0);
        }
    };
    // one k-group: 18 MFMAs on `cur`; the 22 reads of k-group KGN (addresses a_n / b_n) ride in the gaps behind the first eleven; hook(i) is what else rides in gap i
    auto kgroup = [&](auto KGN, Frags& cur, Frags& nxt, unsigned a_n, const unsigned (&b_n)[3], auto&& hook) {
        wp_for<18>([&](auto I) {
            constexpr int i = decltype(I)::value;
            mfma(I, cur);
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (2 * i < 22 && ABL != 2 && ABL != 5) {
                read_op(WIC<2 * i>{}, KGN, nxt, a_n, b_n);
                read_op(WIC<2 * i + 1>{}, KGN, nxt, a_n, b_n);
            }
            hook(I);
            __builtin_amdgcn_sched_barrier(0);
        });
    };
    auto no_hook = [](auto) {};

    // ---- prologue: the x groups behind stage 0 and three stages of requests
    for (int g = wv; g < g0; g += 4) dma_x_group(g);
    dma_stage(0);
    dma_stage(1);
    dma_stage(2);
    bool last8 = rq_mir;                       // requests of the youngest stage in flight: 8 or 6
    wp_wait_vm<12>();                          // everything but (at most) the requests of stages 1 and 2
    __builtin_amdgcn_s_barrier();

    Frags f0, f1;
    memset(&f1, 0, sizeof(f1));
    unsigned a_cur = lane_a + WP_DU0;
    unsigned bq[4][3];                         // the stage's twelve x read addresses
#pragma unroll
    for (int kg = 0; kg < 4; ++kg)
#pragma unroll
        for (int dh = 0; dh < 3; ++dh) bq[kg][dh] = b_addr_of(0, kg, dh);
    wp_for<22>([&](auto R) { read_op(R, WIC<0>{}, f0, a_cur, bq[0]); });
    landed(f0);

    for (int s = 0; s < S; ++s) {
        const int sn = s + 1;
        unsigned bn0[3];                       // k-group 0 of stage s + 1
        kgroup(WIC<1>{}, f0, f1, a_cur, bq[1], no_hook);
        landed(f1);
        kgroup(WIC<2>{}, f1, f0, a_cur, bq[2], no_hook);
        landed(f0);
        kgroup(WIC<3>{}, f0, f1, a_cur, bq[3], [&](auto I) {
            constexpr int i = decltype(I)::value;
            if constexpr (i >= 12 && i < 15) bn0[i - 12] = b_addr_of(sn, 0, i - 12);
        });
        landed(f1);
        if constexpr (ABL != 3 && ABL != 5) {
            if (last8) wp_wait_vm<8>(); else wp_wait_vm<6>();   // stage s + 1 has landed (this wave's share; the barrier makes it everyone's)
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();      // ... and nobody reads stage s any more: its du buffer and the oldest ring rows are free
        }
        __builtin_amdgcn_sched_barrier(0);
        const unsigned a_nx = lane_a + (unsigned)du_stage_off(sn);
        // k-group 3 of stage s | first fragments of stage s + 1, requests of stage s + 3 (sources in gaps 1 and 3, requests in gaps 5 .. 11), the other nine addresses
        kgroup(WIC<0>{}, f1, f0, a_nx, bn0, [&](auto I) {
            constexpr int i = decltype(I)::value;
            if constexpr (ABL != 4 && ABL != 5) {
                if constexpr (i == 1) rq_prep_du(s + 3);
                if constexpr (i == 3) rq_prep_x(4 * (s + 3) + g0 + wv);
                if constexpr (i >= 5 && i < 12 && ABL != 1) rq_issue(WIC<i - 5>{});
            }
            if constexpr (i >= 12 && i < 15) {
#pragma unroll
                for (int dh = 0; dh < 3; ++dh) bq[i - 11][dh] = b_addr_of(sn, i - 11, dh);
            }
        });
        landed(f0);
        last8 = rq_mir;
        a_cur = a_nx;
        bq[0][0] = bn0[0]; bq[0][1] = bn0[1]; bq[0][2] = bn0[2];
    }

    // ---- the accumulators as they are: [wave][a][tap][g][lane] x 4 consecutive filters (D row = filter 8 g + 4 (lane >> 5) + j, column = channel lane & 31)
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");   // the asm MFMAs' results (16 passes) before anything reads their registers
    float* slab = p.part + ((size_t)slice * p.tiles + tile) * WP_SLAB + (size_t)wv * (2 * 9 * 4 * 64 * 4) + lane * 4;
    if constexpr (ABL == 6) {
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int t = 0; t < 9; ++t) asm volatile("" :: "v"(acc[a][t]));
        return;
    }
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int g = 0; g < 4; ++g)
                *(f32x4*)(slab + ((a * 9 + t) * 4 + g) * 256) = f32x4{acc[a][t][4 * g], acc[a][t][4 * g + 1], acc[a][t][4 * g + 2], acc[a][t][4 * g + 3]};
#endif
}

// dW (OIHW fp32) = sum over slices of the slabs.  A unit is 16 bytes of the slab layout (4 consecutive filters of one (tap, channel)): consecutive threads read consecutive
// units of a slice, fully coalesced, up to 8 loads in flight.  SG threads share a unit: thread sg sums the slices [sg per, (sg + 1) per) in order, the partial sums are added
// in sg order through LDS -- a fixed tree, bit-identical from run to run.  (One thread per unit left the 128 -> 256 layers -- 4 tiles, 64 slices: 75 MB behind 73 728 threads --
// latency-bound at 1.7 TB/s: 44 us.)  The 4-byte OIHW writes happen once per element.
template <int SG>
__global__ __launch_bounds__(256) void wgrad_patch_reduce_kernel(const float* __restrict__ part, int tiles, int n_cit, int slices, int cin_real, int cout_real,
                                                                   float* __restrict__ dw) {
    constexpr int UPB = 256 / SG;
    __shared__ f32x4 red[SG > 1 ? SG : 1][UPB];
    const int ul = threadIdx.x % UPB, sg = threadIdx.x / UPB;
    const long long u = (long long)blockIdx.x * UPB + ul;   // 16-byte unit
    const bool live = u < (long long)tiles * (WP_SLAB / 4);
    const int per = (slices + SG - 1) / SG;
    const int s0 = sg * per, s1 = min(s0 + per, slices);
    const size_t stride = (size_t)tiles * WP_SLAB;
    f32x4 s4 = {0.0f, 0.0f, 0.0f, 0.0f};
    if (live) {
        const float* src = part + u * 4;
        int s = s0;
        for (; s + 8 <= s1; s += 8) {
            f32x4 v[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) v[q] = __builtin_nontemporal_load((const f32x4*)(src + (size_t)(s + q) * stride));
#pragma unroll
            for (int q = 0; q < 8; ++q) s4 += v[q];
        }
        if (s < s1) {
            f32x4 v[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) if (s + q < s1) v[q] = __builtin_nontemporal_load((const f32x4*)(src + (size_t)(s + q) * stride));
#pragma unroll
            for (int q = 0; q < 8; ++q) if (s + q < s1) s4 += v[q];
        }
    }
    if constexpr (SG > 1) {
        red[sg][ul] = s4;
        __syncthreads();
        if (sg != 0) return;
#pragma unroll
        for (int q = 1; q < SG; ++q) s4 += red[q][ul];
    }
    if (!live) return;
    const int tile = (int)(u / (WP_SLAB / 4)), r = (int)(u - (long long)tile * (WP_SLAB / 4));
    const int lane = r & 63, g = (r >> 6) & 3, at = r >> 8;          // at = (wave * 2 + a) * 9 + tap
    const int wa = at / 9, tap = at - wa * 9;
    const int a = wa & 1, wv = wa >> 1, wc = wv >> 1, cib = wv & 1;
    const int cot = tile / n_cit, cit = tile - cot * n_cit;
    const int co = cot * 128 + (2 * wc + a) * 32 + 8 * g + 4 * (lane >> 5);
    const int ci = cit * 64 + cib * 32 + (lane & 31);
    if (co >= cout_real || ci >= cin_real) return;
#pragma unroll
    for (int q = 0; q < 4; ++q)
        if (co + q < cout_real) dw[((long long)(co + q) * cin_real + ci) * 9 + tap] = s4[q];
}

struct PatchPlan {
    int tiles, n_cit, slices, per, Qc, back16;
    size_t ws_bytes;
};

// knob "wgrad_patch": 1 the shapes wgrad_big served (>= 16384 pixels); 0 never; 2 every eligible shape (tests)
static bool patch_plan(const y3_conv_desc* d, int n, int h, int w, PatchPlan& pl) {
    const int mode = (int)y3_knob(Y3K_WGRAD_PATCH);
    if (mode == 0 || d->dtype == Y3_F32 || d->ksize != 3 || d->stride != 1) return false;
    if ((d->cin % 64) || (d->cout % 128) || w < 2 || h < 1) return false;
    const int back16 = (w + 3 + 15) / 16 * 16;
    if (back16 > WP_MAX_BACK) return false;
    const long long M = (long long)n * h * w;
    if (mode == 1 && M < 16384) return false;
    const long long Q = ((long long)n * (h + 1) + 1) * (w + 2);
    if (Q >= 0x7fffff00LL) return false;
    const int stages = (int)((Q + WP_KS - 1) / WP_KS);
    pl.tiles = (d->cout / 128) * (d->cin / 64);
    pl.n_cit = d->cin / 64;
    int want = y3_cu_count() / pl.tiles;       // one block per CU
    if (want < 1) want = 1;
    if (want > stages) want = stages;
    const int per_st = (stages + want - 1) / want;
    pl.slices = (stages + per_st - 1) / per_st;
    pl.per = per_st * WP_KS;
    pl.Qc = stages * WP_KS;
    pl.back16 = back16;
    pl.ws_bytes = (size_t)pl.slices * pl.tiles * WP_SLAB * sizeof(float);
    return true;
}

static int launch_patch(const y3_conv_desc* d, const y3_tensor* x, const y3_tensor* du, int cout_real, int cin_real, float* dw, void* ws, const PatchPlan& pl,
                        unsigned x_bytes, unsigned du_bytes, hipStream_t st) {
    PatchArgs a;
    memset(&a, 0, sizeof(a));
    a.x = x->data; a.du = du->data; a.part = (float*)ws;
    a.N = x->n; a.H = x->h; a.W = x->w; a.xpitch = x->pitch; a.dpitch = du->pitch;
    a.x_bytes = x_bytes; a.du_bytes = du_bytes;
    a.PW = x->w + 2; a.PH = x->h + 1;
    a.back16 = pl.back16; a.Qc = pl.Qc; a.per = pl.per; a.tiles = pl.tiles; a.n_cit = pl.n_cit;
    a.dv_pw = y3_make_divisor(a.PW); a.dv_ph = y3_make_divisor(a.PH); a.dv_tiles = y3_make_divisor(pl.tiles); a.dv_cit = y3_make_divisor(pl.n_cit);
    const unsigned blocks = (unsigned)(pl.tiles * pl.slices);
#ifdef Y3_ABLATE
    const char* e = getenv("Y3_WP_ABL");
    const int abl = e ? atoi(e) : 0;
#define Y3_WP_ARM(N) case N: hipLaunchKernelGGL((wgrad_patch_kernel<f16_t, N>), dim3(blocks), dim3(256), 0, st, a); break;
    if (abl && d->dtype == Y3_F16) {
        switch (abl) { Y3_WP_ARM(1) Y3_WP_ARM(2) Y3_WP_ARM(3) Y3_WP_ARM(4) Y3_WP_ARM(5) Y3_WP_ARM(6) Y3_WP_ARM(7) default: break; }
    } else
#endif
    if (d->dtype == Y3_F16) hipLaunchKernelGGL((wgrad_patch_kernel<f16_t>), dim3(blocks), dim3(256), 0, st, a);
    else hipLaunchKernelGGL((wgrad_patch_kernel<bf16_t>), dim3(blocks), dim3(256), 0, st, a);
    Y3_CHECK_LAUNCH();
    const long long units = (long long)pl.tiles * (WP_SLAB / 4);
    // threads per unit: enough of them to keep the slab reads in flight when there are few tiles and many slices
    const int sg = (pl.slices >= 32 && units < 600000) ? 8 : ((pl.slices >= 8 && units < 1200000) ? 4 : 1);
    const unsigned rblocks = (unsigned)((units + 256 / sg - 1) / (256 / sg));
    if (sg == 8) hipLaunchKernelGGL(wgrad_patch_reduce_kernel<8>, dim3(rblocks), dim3(256), 0, st, (const float*)ws, pl.tiles, pl.n_cit, pl.slices, cin_real, cout_real, dw);
    else if (sg == 4) hipLaunchKernelGGL(wgrad_patch_reduce_kernel<4>, dim3(rblocks), dim3(256), 0, st, (const float*)ws, pl.tiles, pl.n_cit, pl.slices, cin_real, cout_real, dw);
    else hipLaunchKernelGGL(wgrad_patch_reduce_kernel<1>, dim3(rblocks), dim3(256), 0, st, (const float*)ws, pl.tiles, pl.n_cit, pl.slices, cin_real, cout_real, dw);
    Y3_CHECK_LAUNCH();
    return 0;
}

}  // namespace
