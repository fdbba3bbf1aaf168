// ComputeLoss (reference utils/loss.py:98-244) for gfx950: build_targets + CIoU box loss + objectness / class
// BCE-with-logits (+ optional focal modulation), forward value and the gradient w.r.t. every prediction tensor.
// Built with -ffp-contract=off.
//
// Data: p[i] is (bs, na, ny, nx, no) of f16/bf16/f32, targets (nt, 6) fp32 [img, cls, x, y, w, h] normalised.
// The reference materialises a match list per level (5 offsets x na anchors x nt targets, filtered) with ~60 tiny
// ATen launches per level; here every POTENTIAL match has a fixed slot  key = (offset*na + anchor)*nt + target
// -- exactly the reference's list order (offset-major, anchor-major, target order; SURVEY 8a' item 12) -- so
//   * no compaction is needed, one wavefront evaluates one slot (lanes parallel over the nc class logits);
//   * "last writer wins" of  tobj[b,a,gj,gi] = iou  (utils/loss.py:161, CPU index_put semantics) becomes
//     atomicMax(winner[cell], key): deterministic and identical to the CPU oracle;
//   * sums are reduced in slot order by a fixed tree -> run-to-run deterministic loss values;
//   * the backward has no floating-point atomics either: every matched slot writes its gradient row into its OWN row of `side`, and the
//     winner slot of a cell that several slots matched adds those rows in ascending slot order (loss_scatter_kernel) -- bit-identical
//     gradients run to run, what the reference gets under torch.use_deterministic_algorithms(True) (train.py:191, utils/general.py:191-205).
// Objectness BCE is one streaming pass over the strided channel-4 plane per level (HBM-bound).
// All math is fp32 on values loaded from p (for f16/bf16 inputs this is at least as accurate as the reference's
// autocast path, whose oracle is the fp32 CPU path anyway); tobj is rounded through p's dtype like the reference.
#include "y3_common.h"

namespace {

constexpr int MAX_NL = 5, MAX_NA = 5;

struct LossDev {
    int nl, na, nc, bs, nt;
    int ny[MAX_NL], nx[MAX_NL];
    float anchors[MAX_NL][MAX_NA][2];
    float balance[MAX_NL];
    float anchor_t, box_gain, obj_gain, cls_gain, cls_pw, obj_pw, cp, cn, fl_gamma;
};

struct LevelWs {
    int* winner;        // cells            (max slot key per cell, -1 = none)
    void* tobj;         // cells of T
    int* slot_cell;     // slots            (-1 = slot not matched)
    float* slot_iou;    // slots            clamp(ciou, 0)
    float* slot_lbox;   // slots            1 - ciou
    float* slot_lcls;   // slots            sum_c BCE(pcls_c, t_c)
    float* side;        // slots * no       fp32 gradient rows, one per matched slot (backward)
    int* slot_dup;      // slots            backward: 1 on a winner slot whose cell is matched by other slots too
    float* obj_part;    // obj_blocks       per-block partial sums of the objectness BCE
    float* sums;        // 4                [n_matches, sum(1-ciou), sum cls bce, sum obj bce]
    long long cells;
    int slots, obj_blocks;
};

Y3_DEV float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }

// BCEWithLogits(pos_weight) (+ FocalLoss wrapper of utils/loss.py:31-63 when gamma > 0): value and d/dx
Y3_DEV void bce_logits(float x, float t, float pw, float gamma, float& loss, float& dldx) {
    const float lw = 1.0f + (pw - 1.0f) * t;
    const float sp = log1pf(expf(-fabsf(x))) + fmaxf(-x, 0.0f);  // softplus(-x)
    const float l = (1.0f - t) * x + lw * sp;
    const float s = sigmoidf_(x);
    const float dl = (1.0f - t) - lw * (1.0f - s);
    if (gamma > 0.0f) {
        const float p_t = t * s + (1.0f - t) * (1.0f - s);
        const float af = t * 0.25f + (1.0f - t) * 0.75f;
        const float om = 1.0f - p_t;
        const float mod = powf(om, gamma);
        const float dpt = (2.0f * t - 1.0f) * s * (1.0f - s);
        const float dmod = om > 0.0f ? -gamma * powf(om, gamma - 1.0f) * dpt : 0.0f;
        loss = l * af * mod;
        dldx = af * (dl * mod + l * dmod);
    } else {
        loss = l;
        dldx = dl;
    }
}

Y3_DEV float tie_lt(float a, float b) { return a < b ? 1.0f : (a == b ? 0.5f : 0.0f); }  // d min(a,b)/da as torch.minimum

// CIoU of a predicted xywh box (from 4 raw logits + anchor) against a target xywh box, and d ciou / d logits.
// Follows upstream bbox_iou(xywh=True, CIoU=True, eps=1e-7) (reference utils/loss.py:148-151); alpha is constant
// under the gradient (torch.no_grad in upstream).
Y3_DEV float ciou_and_grad(const float s[4], float aw, float ah, float tx, float ty, float tw, float th, float ds[4]) {
    const float eps = 1e-7f;
    float sg[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) sg[q] = sigmoidf_(s[q]);
    const float px = sg[0] * 2.0f - 0.5f, py = sg[1] * 2.0f - 0.5f;
    const float d2 = sg[2] * 2.0f, d3 = sg[3] * 2.0f;
    const float pw = d2 * d2 * aw, ph = d3 * d3 * ah;
    const float hw1 = pw / 2.0f, hh1 = ph / 2.0f, hw2 = tw / 2.0f, hh2 = th / 2.0f;
    const float x1 = px - hw1, x2 = px + hw1, y1 = py - hh1, y2 = py + hh1;
    const float X1 = tx - hw2, X2 = tx + hw2, Y1 = ty - hh2, Y2 = ty + hh2;
    const float iw = fminf(x2, X2) - fmaxf(x1, X1), ih = fminf(y2, Y2) - fmaxf(y1, Y1);
    const float iwc = fmaxf(iw, 0.0f), ihc = fmaxf(ih, 0.0f);
    const float inter = iwc * ihc;
    const float uni = pw * ph + tw * th - inter + eps;
    const float iou = inter / uni;
    const float cw = fmaxf(x2, X2) - fminf(x1, X1), ch = fmaxf(y2, Y2) - fminf(y1, Y1);
    const float c2 = cw * cw + ch * ch + eps;
    const float dx = X1 + X2 - x1 - x2, dy = Y1 + Y2 - y1 - y2;
    const float rho2 = (dx * dx + dy * dy) / 4.0f;
    const float A = atanf(tw / th) - atanf(pw / ph);
    const float k4 = 4.0f / (3.14159265358979323846f * 3.14159265358979323846f);
    const float v = k4 * A * A;
    const float alpha = v / (v - iou + (1.0f + eps));
    const float ciou = iou - (rho2 / c2 + v * alpha);
    if (ds) {
        const float g_rho2 = -1.0f / c2, g_c2 = rho2 / (c2 * c2), g_v = -alpha;
        float g_inter = 1.0f / uni;
        const float g_uni = -inter / (uni * uni);
        float g_pw = g_uni * ph, g_ph = g_uni * pw;
        g_inter += -g_uni;
        const float g_iw = g_inter * ihc * (iw >= 0.0f ? 1.0f : 0.0f), g_ih = g_inter * iwc * (ih >= 0.0f ? 1.0f : 0.0f);
        const float g_cw = g_c2 * 2.0f * cw, g_ch = g_c2 * 2.0f * ch;
        const float g_x2 = g_iw * tie_lt(x2, X2) + g_cw * tie_lt(X2, x2) + g_rho2 * (-dx / 2.0f);
        const float g_x1 = -g_iw * tie_lt(X1, x1) - g_cw * tie_lt(x1, X1) + g_rho2 * (-dx / 2.0f);
        const float g_y2 = g_ih * tie_lt(y2, Y2) + g_ch * tie_lt(Y2, y2) + g_rho2 * (-dy / 2.0f);
        const float g_y1 = -g_ih * tie_lt(Y1, y1) - g_ch * tie_lt(y1, Y1) + g_rho2 * (-dy / 2.0f);
        const float r = pw / ph;
        const float dat = 1.0f / (1.0f + r * r);
        const float dv_dpw = k4 * 2.0f * A * (-dat / ph), dv_dph = k4 * 2.0f * A * (dat * pw / (ph * ph));
        g_pw += (g_x2 - g_x1) / 2.0f + g_v * dv_dpw;
        g_ph += (g_y2 - g_y1) / 2.0f + g_v * dv_dph;
        ds[0] = (g_x1 + g_x2) * 2.0f * sg[0] * (1.0f - sg[0]);
        ds[1] = (g_y1 + g_y2) * 2.0f * sg[1] * (1.0f - sg[1]);
        ds[2] = g_pw * 8.0f * sg[2] * sg[2] * (1.0f - sg[2]) * aw;
        ds[3] = g_ph * 8.0f * sg[3] * sg[3] * (1.0f - sg[3]) * ah;
    }
    return ciou;
}

// Slot decode + match test: reference utils/loss.py:208-240 for slot key = (o*na + a)*nt + t.
struct Match {
    bool valid;
    int b, c, a, gi, gj;
    float tx, ty, tw, th, aw, ah;
};
Y3_DEV Match slot_match(const LossDev& P, int lvl, int key, const float* __restrict__ targets) {
    Match m;
    m.valid = false;
    const int nt = P.nt, na = P.na;
    const int t = key % nt;
    const int oa = key / nt;
    const int a = oa % na, o = oa / na;
    const float* tg = targets + (long long)t * 6;
    {   // a target whose image index / class is outside the batch / class range would index out of bounds (the reference raises an
        // IndexError on the host): the slot is dropped here and loss_final_kernel poisons the loss with NaN so the step fails loudly
        const int tb = (int)tg[0], tc = (int)tg[1];
        if (tb < 0 || tb >= P.bs || tc < 0 || tc >= P.nc) return m;
    }
    const float nxf = (float)P.nx[lvl], nyf = (float)P.ny[lvl];
    const float gx = tg[2] * nxf, gy = tg[3] * nyf, gw = tg[4] * nxf, gh = tg[5] * nyf;  // t = targets * gain  (:209-212)
    const float aw = P.anchors[lvl][a][0], ah = P.anchors[lvl][a][1];
    const float rw = gw / aw, rh = gh / ah;
    const float mr = fmaxf(fmaxf(rw, 1.0f / rw), fmaxf(rh, 1.0f / rh));
    if (!(mr < P.anchor_t)) return m;  // (:215-216)
    const float g = 0.5f;
    float offx = 0.0f, offy = 0.0f;
    bool sel = true;
    if (o == 1) { sel = (fmodf(gx, 1.0f) < g) && (gx > 1.0f); offx = g; }            // j: left neighbour   (:221-227)
    else if (o == 2) { sel = (fmodf(gy, 1.0f) < g) && (gy > 1.0f); offy = g; }       // k: upper neighbour
    else if (o == 3) { const float ix = nxf - gx; sel = (fmodf(ix, 1.0f) < g) && (ix > 1.0f); offx = -g; }  // l: right
    else if (o == 4) { const float iy = nyf - gy; sel = (fmodf(iy, 1.0f) < g) && (iy > 1.0f); offy = -g; }  // m: lower
    if (!sel) return m;
    int gi = (int)(gx - offx), gj = (int)(gy - offy);  // .long() truncates toward zero   (:235)
    gi = gi < 0 ? 0 : (gi > P.nx[lvl] - 1 ? P.nx[lvl] - 1 : gi);  // clamp_ (in place: tbox sees the clamped index)  (:239)
    gj = gj < 0 ? 0 : (gj > P.ny[lvl] - 1 ? P.ny[lvl] - 1 : gj);
    m.valid = true;
    m.b = (int)tg[0];
    m.c = (int)tg[1];
    m.a = a;
    m.gi = gi;
    m.gj = gj;
    m.tx = gx - (float)gi;
    m.ty = gy - (float)gj;
    m.tw = gw;
    m.th = gh;
    m.aw = aw;
    m.ah = ah;
    return m;
}

Y3_DEV float wave_sum(float v) {
#pragma unroll
    for (int s = 32; s >= 1; s >>= 1) v += __shfl_xor(v, s);
    return v;
}

// One wavefront per slot.  MODE 0: forward pieces + winner election.  MODE 1: gradient accumulation into side[].
template <typename T, int MODE>
__global__ __launch_bounds__(256) void loss_match_kernel(LossDev P, int lvl, const T* __restrict__ p, const float* __restrict__ targets, LevelWs W,
                                                           const float* __restrict__ scales /* device, [nl][3]; backward only */) {
    const int lane = threadIdx.x & 63;
    const int key = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (key >= W.slots) return;
    const Match m = slot_match(P, lvl, key, targets);
    if (!m.valid) {
        if (MODE == 0 && lane == 0) W.slot_cell[key] = -1;
        return;
    }
    const int no = P.nc + 5;
    const long long cell = (((long long)m.b * P.na + m.a) * P.ny[lvl] + m.gj) * P.nx[lvl] + m.gi;
    const T* __restrict__ row = p + cell * no;
    float s[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) s[q] = to_f32<T>(row[q]);
    float ds[4];
    const float ciou = ciou_and_grad(s, m.aw, m.ah, m.tx, m.ty, m.tw, m.th, MODE == 1 ? ds : nullptr);
    if (MODE == 0) {
        float lc = 0.0f;
        if (P.nc > 1) {  // class BCE only with several classes  (:164)
            for (int c = lane; c < P.nc; c += 64) {
                float l, d;
                bce_logits(to_f32<T>(row[5 + c]), c == m.c ? P.cp : P.cn, P.cls_pw, P.fl_gamma, l, d);
                lc += l;
            }
            lc = wave_sum(lc);
        }
        if (lane == 0) {
            W.slot_cell[key] = (int)cell;
            W.slot_iou[key] = fmaxf(ciou, 0.0f);  // iou.detach().clamp(0)   (:155)
            W.slot_lbox[key] = 1.0f - ciou;
            W.slot_lcls[key] = lc;
            atomicMax(&W.winner[cell], key);
        }
    } else {
        // d loss / d row of THIS slot into its own row of `side` (plain stores); a slot that shares its cell with the winner flags the winner, whose
        // scatter pass then adds the rows of the cell in slot order (duplicates sum, as autograd's index backward does -- in a fixed order)
        const float scale_box = scales[lvl * 3 + 0], scale_cls = scales[lvl * 3 + 1];
        const int owner = W.winner[cell];
        float* __restrict__ acc = W.side + (long long)key * no;
        if (lane < 4) acc[lane] = -scale_box * ds[lane];  // lbox = mean(1 - ciou)
        for (int c = lane; c < P.nc; c += 64) {
            float l, d = 0.0f;
            if (P.nc > 1) bce_logits(to_f32<T>(row[5 + c]), c == m.c ? P.cp : P.cn, P.cls_pw, P.fl_gamma, l, d);   // (one class: no class loss, the row's entry is 0)
            acc[5 + c] = scale_cls * d;
        }
        if (lane == 0 && owner != key) W.slot_dup[owner] = 1;   // (every writer stores the same value)
    }
}

// ComputeLoss.sort_obj_iou (utils/loss.py:156-158): the matches are permuted into ascending-iou order before `tobj[b, a, gj, gi] = iou`, so a cell matched several times
// keeps its LARGEST iou.  The winner slot of a cell stays the election above (it only names where the cell's gradient rows accumulate); every other slot of the cell raises
// the winner's iou to its own -- non-negative floats order like their bit patterns, so an integer atomicMax is exact and independent of the order the slots arrive in.
__global__ void loss_cell_max_iou_kernel(LevelWs W) {
    const int key = blockIdx.x * 256 + threadIdx.x;
    if (key >= W.slots) return;
    const int cell = W.slot_cell[key];
    if (cell < 0) return;
    const int owner = W.winner[cell];
    if (owner != key) atomicMax((int*)&W.slot_iou[owner], __float_as_int(W.slot_iou[key]));
}

template <typename T> __global__ void loss_tobj_kernel(LevelWs W) {
    const int key = blockIdx.x * 256 + threadIdx.x;
    if (key >= W.slots) return;
    const int cell = W.slot_cell[key];
    if (cell >= 0 && W.winner[cell] == key) ((T*)W.tobj)[cell] = from_f32<T>(W.slot_iou[key]);
}

// Objectness BCE over every cell of one level.  MODE 0: per-block partial sums.  MODE 1: writes the WHOLE gradient
// tensor of the level: zeros everywhere except channel 4 (matched cells' other channels are filled afterwards).
template <typename T, int MODE>
__global__ __launch_bounds__(256) void loss_obj_kernel(LossDev P, int lvl, const T* __restrict__ p, LevelWs W, T* __restrict__ gp,
                                                         const float* __restrict__ scales) {
    const int no = P.nc + 5;
    if (MODE == 0) {
        __shared__ float red[4];
        const long long cell = (long long)blockIdx.x * 256 + threadIdx.x;
        float l = 0.0f;
        if (cell < W.cells) {
            float d;
            bce_logits(to_f32<T>(p[cell * no + 4]), to_f32<T>(((const T*)W.tobj)[cell]), P.obj_pw, P.fl_gamma, l, d);
        }
        l = wave_sum(l);
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = l;
        __syncthreads();
        if (threadIdx.x == 0) W.obj_part[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
    } else {
        // one 16-byte store per thread (V consecutive elements, the (cell, channel) cursor advanced by hand): a thread per ELEMENT -- a
        // 64-bit division and a 2-byte store each -- wrote the 209 MB of the 80x80 level at batch 64 at 0.44 TB/s (470 us)
        constexpr int V = 16 / (int)sizeof(T);
        const long long total = W.cells * no;
        const long long i0 = ((long long)blockIdx.x * 256 + threadIdx.x) * V;
        if (i0 >= total) return;
        long long cell = i0 / no;
        int ch = (int)(i0 - cell * no);
        const float sc = scales[lvl * 3 + 2];
        T o[V];
        const int nv = total - i0 < V ? (int)(total - i0) : V;
#pragma unroll
        for (int q = 0; q < V; ++q) o[q] = from_f32<T>(0.0f);
        // the objectness elements of this run sit at q4, q4 + no, ... (at most two in 8 elements: no >= 6): their loads are issued
        // together, outside any per-element branch (a divergent load per element serialised up to 8 memory round trips per wave)
        int q4 = 4 - ch;
        if (q4 < 0) q4 += no;
        for (int q = q4; q < nv; q += no) {
            const int wrap = ch + q;
            const long long cq = cell + (wrap >= 2 * no ? 2 : (wrap >= no ? 1 : 0));
            float l, d;
            bce_logits(to_f32<T>(p[i0 + q]), to_f32<T>(((const T*)W.tobj)[cq]), P.obj_pw, P.fl_gamma, l, d);
            const T g = from_f32<T>(d * sc);
#pragma unroll
            for (int k = 0; k < V; ++k) if (k == q) o[k] = g;
        }
        if (nv == V && ((uintptr_t)gp & 15) == 0) {
            typedef unsigned int u4 __attribute__((ext_vector_type(4)));
            u4 raw;
            __builtin_memcpy(&raw, o, 16);
            *(u4*)(gp + i0) = raw;
        } else {
            for (int q = 0; q < nv; ++q) gp[i0 + q] = o[q];
        }
    }
}

// matched cells: gradient rows from the fp32 slot rows (winner slots only), channel 4 untouched.  A cell matched once takes its row as it is.  A cell matched
// several times (flagged by loss_match_kernel<1>) sums the rows of ALL its slots in ascending slot order: the slots of cell (b, a, gj, gi) can only be the keys
// (o * na + a) * nt + t, so the winner's wave walks those 5 * nt candidates 64 at a time and adds the rows of the ones whose slot_cell is this cell -- a fixed
// order, no atomics, bit-identical from run to run.
template <typename T> __global__ __launch_bounds__(256) void loss_scatter_kernel(LossDev P, LevelWs W, T* __restrict__ gp) {
    const int lane = threadIdx.x & 63;
    const int key = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (key >= W.slots) return;
    const int cell = W.slot_cell[key];
    if (cell < 0 || W.winner[cell] != key) return;
    const int no = P.nc + 5;
    T* __restrict__ row = gp + (long long)cell * no;
    if (!W.slot_dup[key]) {
        const float* __restrict__ acc = W.side + (long long)key * no;
        for (int ch = lane; ch < no; ch += 64)
            if (ch != 4) row[ch] = from_f32<T>(acc[ch]);
        return;
    }
    const int nt = P.nt, na = P.na;
    const int a = (key / nt) % na;
    for (int ch0 = 0; ch0 < no; ch0 += 64) {
        const int ch = ch0 + lane;
        float sum = 0.0f;
        for (int o = 0; o < 5; ++o) {
            const int kbase = (o * na + a) * nt;
            for (int t0 = 0; t0 < nt; t0 += 64) {
                const int t = t0 + lane;
                unsigned long long hit = __ballot(t < nt && W.slot_cell[kbase + t] == cell);
                while (hit) {   // (wave-uniform: ascending target index = ascending slot key)
                    const int j = __builtin_ctzll(hit);
                    hit &= hit - 1;
                    if (ch < no) sum += W.side[(long long)(kbase + t0 + j) * no + ch];
                }
            }
        }
        if (ch < no && ch != 4) row[ch] = from_f32<T>(sum);
    }
}

// deterministic single-block reductions: slot arrays (in slot order) and objectness partials
__global__ __launch_bounds__(256) void loss_reduce_level_kernel(LevelWs W) {
    __shared__ float sm[3][256];
    __shared__ float so[256];
    const int tid = threadIdx.x;
    float n = 0.0f, lb = 0.0f, lc = 0.0f, lo = 0.0f;
    for (int k = tid; k < W.slots; k += 256)
        if (W.slot_cell[k] >= 0) { n += 1.0f; lb += W.slot_lbox[k]; lc += W.slot_lcls[k]; }
    for (int k = tid; k < W.obj_blocks; k += 256) lo += W.obj_part[k];
    sm[0][tid] = n; sm[1][tid] = lb; sm[2][tid] = lc; so[tid] = lo;
    __syncthreads();
    for (int s = 128; s >= 1; s >>= 1) {
        if (tid < s) { sm[0][tid] += sm[0][tid + s]; sm[1][tid] += sm[1][tid + s]; sm[2][tid] += sm[2][tid + s]; so[tid] += so[tid + s]; }
        __syncthreads();
    }
    if (tid == 0) { W.sums[0] = sm[0][0]; W.sums[1] = sm[1][0]; W.sums[2] = sm[2][0]; W.sums[3] = so[0]; }
}

struct FinalArgs { const float* sums[MAX_NL]; long long cells[MAX_NL]; };
__global__ void loss_final_kernel(LossDev P, FinalArgs F, const float* __restrict__ targets, float* __restrict__ out) {
    bool bad = false;   // one wave scans the target list for rows slot_match had to drop
    for (int t = threadIdx.x; t < P.nt; t += 64) {
        const int tb = (int)targets[(long long)t * 6], tc = (int)targets[(long long)t * 6 + 1];
        bad |= tb < 0 || tb >= P.bs || tc < 0 || tc >= P.nc;
    }
    bad = __ballot(bad) != 0ull;
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    if (bad) {
        const float qnan = __builtin_nanf("");
        out[0] = out[1] = out[2] = out[3] = qnan;
        return;
    }
    float lbox = 0.0f, lobj = 0.0f, lcls = 0.0f;
    for (int i = 0; i < P.nl; ++i) {
        const float n = F.sums[i][0];
        if (n > 0.0f) {
            lbox += F.sums[i][1] / n;                                   // (1 - iou).mean()          (:152)
            if (P.nc > 1) lcls += F.sums[i][2] / (n * (float)P.nc);     // BCEcls mean over n*nc     (:167)
        }
        lobj += (F.sums[i][3] / (float)F.cells[i]) * P.balance[i];      // BCEobj mean * balance     (:169-170)
    }
    lbox *= P.box_gain; lobj *= P.obj_gain; lcls *= P.cls_gain;        // (:176-178)
    out[0] = (lbox + lobj + lcls) * (float)P.bs;                        // (:181)
    out[1] = lbox; out[2] = lobj; out[3] = lcls;
}

size_t al(size_t v) { return (v + 255) & ~(size_t)255; }

size_t carve_level(LevelWs& W, unsigned char* base, size_t off, const y3_loss_params* p, int lvl, int nt, int esz) {
    auto take = [&](size_t bytes) { unsigned char* q = base ? base + off : nullptr; off += al(bytes); return q; };
    const int no = p->nc + 5;
    W.cells = (long long)p->bs * p->na * p->ny[lvl] * p->nx[lvl];
    W.slots = 5 * p->na * nt;
    W.obj_blocks = (int)((W.cells + 255) / 256);
    W.winner = (int*)take((size_t)W.cells * 4);
    W.tobj = take((size_t)W.cells * esz);
    W.slot_cell = (int*)take((size_t)(W.slots + 1) * 4);
    W.slot_iou = (float*)take((size_t)(W.slots + 1) * 4);
    W.slot_lbox = (float*)take((size_t)(W.slots + 1) * 4);
    W.slot_lcls = (float*)take((size_t)(W.slots + 1) * 4);
    W.side = (float*)take((size_t)(W.slots + 1) * no * 4);
    W.slot_dup = (int*)take((size_t)(W.slots + 1) * 4);
    W.obj_part = (float*)take((size_t)W.obj_blocks * 4);
    W.sums = (float*)take(16);
    return off;
}

int check_params(const y3_loss_params* p, int nt) {
    if (!p) Y3_FAIL("y3_loss: null params");
    if (p->nl < 1 || p->nl > MAX_NL || p->na < 1 || p->na > MAX_NA) Y3_FAIL("y3_loss: nl=%d na=%d unsupported", p->nl, p->na);
    if (p->nc < 1 || p->bs < 1 || nt < 0) Y3_FAIL("y3_loss: bad nc/bs/nt");
    for (int i = 0; i < p->nl; ++i) {
        const long long cells = (long long)p->bs * p->na * p->ny[i] * p->nx[i];
        if (cells <= 0 || cells * (p->nc + 5) > 0x7fffffffLL) Y3_FAIL("y3_loss: level %d too large", i);
    }
    if ((long long)5 * p->na * nt > 0x3fffffffLL) Y3_FAIL("y3_loss: too many targets");
    return 0;
}

LossDev to_dev(const y3_loss_params* p, int nt) {
    LossDev d;
    memset(&d, 0, sizeof(d));
    d.nl = p->nl; d.na = p->na; d.nc = p->nc; d.bs = p->bs; d.nt = nt;
    for (int i = 0; i < p->nl; ++i) {
        d.ny[i] = p->ny[i]; d.nx[i] = p->nx[i]; d.balance[i] = p->balance[i];
        for (int a = 0; a < p->na; ++a) { d.anchors[i][a][0] = p->anchors[(i * p->na + a) * 2]; d.anchors[i][a][1] = p->anchors[(i * p->na + a) * 2 + 1]; }
    }
    d.anchor_t = p->anchor_t; d.box_gain = p->box_gain; d.obj_gain = p->obj_gain; d.cls_gain = p->cls_gain;
    d.cls_pw = p->cls_pw; d.obj_pw = p->obj_pw; d.cp = p->cp; d.cn = p->cn; d.fl_gamma = p->fl_gamma;
    return d;
}

template <typename T>
int loss_fwd(const y3_loss_params* p, const void* const* preds, const float* targets, int nt, float* out4, unsigned char* ws, hipStream_t st) {
    const LossDev D = to_dev(p, nt);
    FinalArgs F;
    memset(&F, 0, sizeof(F));
    size_t off = 0;
    for (int i = 0; i < p->nl; ++i) {
        LevelWs W;
        off = carve_level(W, ws, off, p, i, nt, sizeof(T));
        F.sums[i] = W.sums;
        F.cells[i] = W.cells;
        Y3_HIP(hipMemsetAsync(W.winner, 0xff, (size_t)W.cells * 4, st));
        Y3_HIP(hipMemsetAsync(W.tobj, 0, (size_t)W.cells * sizeof(T), st));
        if (W.slots > 0) {
            hipLaunchKernelGGL((loss_match_kernel<T, 0>), dim3((W.slots + 3) / 4), dim3(256), 0, st, D, i, (const T*)preds[i], targets, W, (const float*)nullptr);
            Y3_CHECK_LAUNCH();
            if (p->sort_obj_iou) {
                hipLaunchKernelGGL(loss_cell_max_iou_kernel, dim3((W.slots + 255) / 256), dim3(256), 0, st, W);
                Y3_CHECK_LAUNCH();
            }
            hipLaunchKernelGGL((loss_tobj_kernel<T>), dim3((W.slots + 255) / 256), dim3(256), 0, st, W);
            Y3_CHECK_LAUNCH();
        }
        hipLaunchKernelGGL((loss_obj_kernel<T, 0>), dim3(W.obj_blocks), dim3(256), 0, st, D, i, (const T*)preds[i], W, (T*)nullptr, (const float*)nullptr);
        Y3_CHECK_LAUNCH();
        hipLaunchKernelGGL(loss_reduce_level_kernel, dim3(1), dim3(256), 0, st, W);
        Y3_CHECK_LAUNCH();
    }
    hipLaunchKernelGGL(loss_final_kernel, dim3(1), dim3(64), 0, st, D, F, targets, out4);
    Y3_CHECK_LAUNCH();
    return 0;
}

// scale factors need the per-level match count, which lives on the device: a tiny kernel turns it into the
// three per-level multipliers so the backward needs no host round trip.
__global__ void loss_scales_kernel(LossDev P, FinalArgs F, const float* __restrict__ targets, const float* __restrict__ grad_out, float* __restrict__ scales /* nl*3 */) {
    // a target row outside the batch / class range made the forward return NaN (loss_final_kernel; the reference raises IndexError on the
    // host); the BACKWARD of such a call is poisoned the same way -- every gradient NaN, so GradScaler's found-inf skips the step
    // (round-2 advisor finding: finite gradients of the valid rows used to let the optimizer step on a batch the reference would have rejected)
    bool bad = false;
    for (int t = threadIdx.x; t < P.nt; t += 64) {
        const int tb = (int)targets[(long long)t * 6], tc = (int)targets[(long long)t * 6 + 1];
        bad |= tb < 0 || tb >= P.bs || tc < 0 || tc >= P.nc;
    }
    bad = __ballot(bad) != 0ull;
    const int i = threadIdx.x;
    if (i >= P.nl) return;
    const float go = bad ? __builtin_nanf("") : (grad_out ? grad_out[0] : 1.0f);
    const float n = F.sums[i][0];
    const float bsf = (float)P.bs;
    scales[i * 3 + 0] = (n > 0.0f || bad) ? go * bsf * P.box_gain / (n > 0.0f ? n : 1.0f) : 0.0f;
    scales[i * 3 + 1] = ((n > 0.0f && P.nc > 1) || bad) ? go * bsf * P.cls_gain / ((n > 0.0f ? n : 1.0f) * (float)P.nc) : 0.0f;
    scales[i * 3 + 2] = go * bsf * P.obj_gain * P.balance[i] / (float)F.cells[i];
}

template <typename T>
int loss_bwd(const y3_loss_params* p, const void* const* preds, const float* targets, int nt, const float* grad_out, void* const* grads, unsigned char* ws,
             hipStream_t st) {
    const LossDev D = to_dev(p, nt);
    FinalArgs F;
    memset(&F, 0, sizeof(F));
    LevelWs Ws[MAX_NL];
    size_t off = 0;
    for (int i = 0; i < p->nl; ++i) {
        off = carve_level(Ws[i], ws, off, p, i, nt, sizeof(T));
        F.sums[i] = Ws[i].sums;
        F.cells[i] = Ws[i].cells;
    }
    float* scales = (float*)(ws + off);
    hipLaunchKernelGGL(loss_scales_kernel, dim3(1), dim3(64), 0, st, D, F, targets, grad_out, scales);
    Y3_CHECK_LAUNCH();
    const int no = p->nc + 5;
    for (int i = 0; i < p->nl; ++i) {
        LevelWs& W = Ws[i];
        const long long elems = W.cells * no;
        const long long per_block = 256LL * (16 / (long long)sizeof(T));   // 16 bytes of the gradient per thread
        hipLaunchKernelGGL((loss_obj_kernel<T, 1>), dim3((unsigned)((elems + per_block - 1) / per_block)), dim3(256), 0, st, D, i, (const T*)preds[i], W, (T*)grads[i], scales);
        Y3_CHECK_LAUNCH();
        if (W.slots > 0) {
            Y3_HIP(hipMemsetAsync(W.slot_dup, 0, (size_t)W.slots * 4, st));   // (the slot rows need no zero fill: every matched slot writes its whole row)
            hipLaunchKernelGGL((loss_match_kernel<T, 1>), dim3((W.slots + 3) / 4), dim3(256), 0, st, D, i, (const T*)preds[i], targets, W, (const float*)scales);
            Y3_CHECK_LAUNCH();
            hipLaunchKernelGGL((loss_scatter_kernel<T>), dim3((W.slots + 3) / 4), dim3(256), 0, st, D, W, (T*)grads[i]);
            Y3_CHECK_LAUNCH();
        }
    }
    return 0;
}

size_t total_ws(const y3_loss_params* p, int nt, int esz) {
    size_t off = 0;
    LevelWs W;
    for (int i = 0; i < p->nl; ++i) off = carve_level(W, nullptr, off, p, i, nt, esz);
    return off + al(MAX_NL * 3 * sizeof(float));
}

}  // namespace

extern "C" size_t y3_loss_workspace_bytes(const y3_loss_params* p, int32_t nt) {
    if (check_params(p, nt) != 0) return 0;
    return total_ws(p, nt, 4);
}

// per-level objectness loss (the mean BCE the forward summed per level) out of the workspace a y3_loss_fwd call filled: what
// ComputeLoss(autobalance=True) reads back per step (reference utils/loss.py:171-175, `obji.detach().item()`)
__global__ void loss_level_obj_kernel(FinalArgs F, int nl, float* __restrict__ out) {
    const int i = threadIdx.x;
    if (i < nl) out[i] = F.sums[i][3] / (float)F.cells[i];
}
extern "C" int y3_loss_level_obj(const y3_loss_params* p, int32_t dtype, int32_t nt, void* workspace, size_t workspace_bytes, float* obj_levels, void* stream) {
    if (check_params(p, nt) != 0) return -1;
    if (!workspace || !obj_levels) Y3_FAIL("y3_loss_level_obj: null argument");
    if (dtype != Y3_F16 && dtype != Y3_BF16 && dtype != Y3_F32) Y3_FAIL("y3_loss_level_obj: bad dtype %d", dtype);
    const int esz = dtype == Y3_F32 ? 4 : 2;   // the workspace is carved with the element size of the predictions (the tobj planes)
    if (workspace_bytes < total_ws(p, nt, 4)) Y3_FAIL("y3_loss_level_obj: workspace too small");
    FinalArgs F;
    size_t off = 0;
    for (int i = 0; i < p->nl; ++i) {
        LevelWs W;
        off = carve_level(W, (unsigned char*)workspace, off, p, i, nt, esz);
        F.sums[i] = W.sums;
        F.cells[i] = W.cells;
    }
    hipLaunchKernelGGL(loss_level_obj_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, F, p->nl, obj_levels);
    Y3_CHECK_LAUNCH();
    return 0;
}

extern "C" int y3_loss_fwd(const y3_loss_params* p, int32_t dtype, const void* const* preds, const float* targets, int32_t nt, float* out4, void* workspace,
                           size_t workspace_bytes, void* stream) {
    if (check_params(p, nt) != 0) return -1;
    if (!preds || !out4 || !workspace || (nt > 0 && !targets)) Y3_FAIL("y3_loss_fwd: null argument");
    if (workspace_bytes < total_ws(p, nt, 4)) Y3_FAIL("y3_loss_fwd: workspace too small");
    hipStream_t st = (hipStream_t)stream;
    switch (dtype) {
        case Y3_F16: return loss_fwd<f16_t>(p, preds, targets, nt, out4, (unsigned char*)workspace, st);
        case Y3_BF16: return loss_fwd<bf16_t>(p, preds, targets, nt, out4, (unsigned char*)workspace, st);
        case Y3_F32: return loss_fwd<float>(p, preds, targets, nt, out4, (unsigned char*)workspace, st);
    }
    Y3_FAIL("y3_loss_fwd: bad dtype %d", dtype);
}

extern "C" int y3_loss_bwd(const y3_loss_params* p, int32_t dtype, const void* const* preds, const float* targets, int32_t nt, const float* grad_out,
                           void* const* grads, void* workspace, size_t workspace_bytes, void* stream) {
    if (check_params(p, nt) != 0) return -1;
    if (!preds || !grads || !workspace || (nt > 0 && !targets)) Y3_FAIL("y3_loss_bwd: null argument");
    if (workspace_bytes < total_ws(p, nt, 4)) Y3_FAIL("y3_loss_bwd: workspace too small");
    hipStream_t st = (hipStream_t)stream;
    switch (dtype) {
        case Y3_F16: return loss_bwd<f16_t>(p, preds, targets, nt, grad_out, grads, (unsigned char*)workspace, st);
        case Y3_BF16: return loss_bwd<bf16_t>(p, preds, targets, nt, grad_out, grads, (unsigned char*)workspace, st);
        case Y3_F32: return loss_bwd<float>(p, preds, targets, nt, grad_out, grads, (unsigned char*)workspace, st);
    }
    Y3_FAIL("y3_loss_bwd: bad dtype %d", dtype);
}
