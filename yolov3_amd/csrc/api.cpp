// Error plumbing, version and the run-time tuning table of libyolov3_hip.so.
#include <atomic>
#include <mutex>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../../include/yolov3_hip.h"
#include "y3_knobs.h"

static thread_local char g_err[512] = "";

void y3_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char* y3_last_error(void) { return g_err; }
extern "C" int y3_abi_version(void) { return Y3_ABI_VERSION; }

// ---- tuning table (ids: y3_common.h::y3_knob_id, same order) ---------------------------------
namespace {
struct Knob {
    const char* name;
    long long def;
};
const Knob kKnobs[] = {
    {"conv", 0}, {"conv_ahead", 3}, {"bn_nt_bytes", 64ll << 20},
    {"wgrad", 0}, {"wgrad_xcd", 2}, {"dgrad_quad", 1}, {"spp_direct", 0}, {"wgrad_strip", 1}, {"conv_strip", 1},
    {"conv_v10", 1}, {"v10_mp", 0}, {"v10_blocks", 0}, {"v10_half", 2}, {"v10_ksplit", 1}, {"v10_slices", 0},
    {"v10_group", 1}, {"tile_xcd", 1}, {"conv_1x1s", 1}, {"wgrad_patch", 1}, {"wgrad_blocks", 512}, {"nms_sort", 1}, {"v10_defer", 0},
};
constexpr int kCount = (int)(sizeof(kKnobs) / sizeof(kKnobs[0]));
static_assert(kCount == Y3K_COUNT, "kKnobs and y3_knob_id (y3_common.h) list the same knobs in the same order");
std::atomic<long long> g_val[kCount];
std::once_flag g_once;

int knob_index(const char* key) {
    for (int i = 0; i < kCount; ++i)
        if (key && !strcmp(key, kKnobs[i].name)) return i;
    return -1;
}
void load_defaults() {
    for (int i = 0; i < kCount; ++i) g_val[i].store(kKnobs[i].def, std::memory_order_relaxed);
    const char* e = getenv("Y3_TUNE");   // "key=value,key=value": read here and nowhere else
    if (!e) return;
    char buf[512];
    strncpy(buf, e, sizeof(buf) - 1);
    buf[sizeof(buf) - 1] = 0;
    for (char* tok = strtok(buf, ",;"); tok; tok = strtok(nullptr, ",;")) {
        char* eq = strchr(tok, '=');
        if (!eq) continue;
        *eq = 0;
        const int i = knob_index(tok);
        if (i >= 0) g_val[i].store(atoll(eq + 1), std::memory_order_relaxed);
        else fprintf(stderr, "libyolov3_hip: Y3_TUNE names an unknown knob '%s'\n", tok);
    }
}
}  // namespace

long long y3_knob(int id) {
    std::call_once(g_once, load_defaults);
    return (id >= 0 && id < kCount) ? g_val[id].load(std::memory_order_relaxed) : 0;
}

extern "C" int y3_tune_set(const char* key, int64_t value) {
    std::call_once(g_once, load_defaults);
    const int i = knob_index(key);
    if (i < 0) {
        y3_set_error("y3_tune_set: unknown knob '%s'", key ? key : "(null)");
        return -1;
    }
    g_val[i].store((long long)value, std::memory_order_relaxed);
    return 0;
}

extern "C" int64_t y3_tune_get(const char* key) {
    std::call_once(g_once, load_defaults);
    const int i = knob_index(key);
    if (i < 0) {
        y3_set_error("y3_tune_get: unknown knob '%s'", key ? key : "(null)");
        return INT64_MIN;
    }
    return (int64_t)g_val[i].load(std::memory_order_relaxed);
}

extern "C" void y3_tune_reset(void) {
    std::call_once(g_once, load_defaults);
    load_defaults();
}
