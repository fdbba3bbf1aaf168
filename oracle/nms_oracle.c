/* Greedy NMS in plain C -- TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).
 *
 * Restates the published CPU algorithm of torchvision.ops.nms (torchvision>=0.9.0,
 * reference requirements.txt:17; un-vendored), the single native op the reference's
 * non_max_suppression delegates to (reference utils/general.py:733):
 *   - areas = (x2-x1)*(y2-y1) in float32
 *   - visit boxes by STABLE descending score order
 *   - a visited, un-suppressed box i is kept; every later j with
 *       inter / (area_i + area_j - inter) > thr      (strict; float32 ratio compared with the DOUBLE threshold)
 *     is suppressed, where inter = max(0, min(x2)-max(x1)) * max(0, min(y2)-max(y1))
 *   - returns kept indices in visiting order.
 * Build with -ffp-contract=off so no product is fused into the adds.
 */
#include <stdlib.h>
#include <string.h>

static void merge_sort_desc(const float* key, long* idx, long* tmp, long n) {
    /* bottom-up stable merge sort of idx by key descending */
    for (long width = 1; width < n; width *= 2) {
        for (long lo = 0; lo < n; lo += 2 * width) {
            long mid = lo + width < n ? lo + width : n;
            long hi = lo + 2 * width < n ? lo + 2 * width : n;
            long a = lo, b = mid, o = lo;
            while (a < mid && b < hi) {
                /* take from the right run only when strictly greater: keeps ties stable */
                if (key[idx[b]] > key[idx[a]]) tmp[o++] = idx[b++];
                else tmp[o++] = idx[a++];
            }
            while (a < mid) tmp[o++] = idx[a++];
            while (b < hi) tmp[o++] = idx[b++];
        }
        memcpy(idx, tmp, (size_t)n * sizeof(long));
    }
}

long y3o_nms(const float* boxes, const float* scores, long n, double thr, long* keep_out) {
    if (n <= 0) return 0;
    long* order = (long*)malloc((size_t)n * sizeof(long));
    long* tmp = (long*)malloc((size_t)n * sizeof(long));
    float* area = (float*)malloc((size_t)n * sizeof(float));
    unsigned char* dead = (unsigned char*)calloc((size_t)n, 1);
    for (long i = 0; i < n; ++i) {
        order[i] = i;
        const float* b = boxes + 4 * i;
        area[i] = (b[2] - b[0]) * (b[3] - b[1]);
    }
    merge_sort_desc(scores, order, tmp, n);
    long nk = 0;
    for (long a = 0; a < n; ++a) {
        long i = order[a];
        if (dead[i]) continue;
        keep_out[nk++] = i;
        const float ix1 = boxes[4 * i], iy1 = boxes[4 * i + 1], ix2 = boxes[4 * i + 2], iy2 = boxes[4 * i + 3];
        const float iarea = area[i];
        for (long c = a + 1; c < n; ++c) {
            long j = order[c];
            if (dead[j]) continue;
            const float* b = boxes + 4 * j;
            float xx1 = ix1 > b[0] ? ix1 : b[0];
            float yy1 = iy1 > b[1] ? iy1 : b[1];
            float xx2 = ix2 < b[2] ? ix2 : b[2];
            float yy2 = iy2 < b[3] ? iy2 : b[3];
            float w = xx2 - xx1; if (!(w > 0.0f)) w = 0.0f;
            float h = yy2 - yy1; if (!(h > 0.0f)) h = 0.0f;
            float inter = w * h;
            float ovr = inter / (iarea + area[j] - inter);
            if ((double)ovr > thr) dead[j] = 1; /* float ovr vs double threshold, as torchvision's nms_kernel_impl */
        }
    }
    free(order); free(tmp); free(area); free(dead);
    return nk;
}
