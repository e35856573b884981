// Host-side runtime of the step: events and cross-stream edges driven through the library, and a launch tape that
// replays one iteration's enqueue calls from C.
//
// Why events live here: a step forks and joins four streams about sixty times.  Each edge is a record on one stream and
// a wait on another; done through a framework's stream objects every edge creates, records and destroys an event of its
// own, also while a graph capture is open.  Here every event is created once, before anything is recorded, and an edge
// is two runtime calls on existing objects: nothing is created or destroyed between pgan_step_graph_begin and _end
// (pgan_events_created lets a caller assert that).
//
// Why the tape lives here: an iteration is ~700 enqueue calls whose arguments (raw pointers, sizes, stream handles)
// never change.  Noted once while an iteration runs, they are replayed by one call that walks a flat array -- the same
// launches on the same streams in the same order, with no interpreter between them.  The reference launches its step
// op by op from Python (src/train_wgan_v2.py:516-634).
#include <atomic>
#include <cstring>
#include <utility>
#include <vector>

#include "../../include/pgan_hip.h"
#include "pgan_common.h"

namespace {

std::atomic<long long> g_events_created{0};

// ---- calling an entry point from an array of 64-bit words -------------------------------------------------------
// One word per argument: integers sign-extended, pointers as addresses, floats as their 32 bits in the low half.
template <class T>
inline T word_as(uint64_t w) {
    if constexpr (std::is_pointer_v<T>) {
        return reinterpret_cast<T>(static_cast<uintptr_t>(w));
    } else if constexpr (std::is_same_v<T, float>) {
        const uint32_t b = static_cast<uint32_t>(w);
        float f;
        std::memcpy(&f, &b, 4);
        return f;
    } else {
        static_assert(std::is_integral_v<T>, "argument kind");
        return static_cast<T>(static_cast<int64_t>(w));
    }
}
template <class... A, size_t... I>
inline int call_words(int (*fn)(A...), const uint64_t* w, std::index_sequence<I...>) {
    return fn(word_as<A>(w[I])...);
}
template <auto FN>
struct Thunk;
template <class... A, int (*FN)(A...)>
struct Thunk<FN> {
    static constexpr int nargs = (int)sizeof...(A);
    static int call(const uint64_t* w) { return call_words(FN, w, std::index_sequence_for<A...>{}); }
};

struct Entry {
    const char* name;
    int (*call)(const uint64_t*);
    int nargs;
};
#define PGAN_ENTRY(f) {#f, &Thunk<&f>::call, Thunk<&f>::nargs}
// every entry point that only enqueues work on a stream (queries, host parsers and graph management are not replayable)
const Entry ENTRIES[] = {
    PGAN_ENTRY(pgan_conv_down), PGAN_ENTRY(pgan_conv_down_gn_sums), PGAN_ENTRY(pgan_conv_up), PGAN_ENTRY(pgan_conv_wgrad),
    PGAN_ENTRY(pgan_rows_reduce), PGAN_ENTRY(pgan_rows_reduce_multi),
    PGAN_ENTRY(pgan_conv_x3_prep), PGAN_ENTRY(pgan_conv_x3_prep_multi), PGAN_ENTRY(pgan_conv_down_x3),
    PGAN_ENTRY(pgan_conv_up_x3), PGAN_ENTRY(pgan_conv_wgrad_x3), PGAN_ENTRY(pgan_conv_down_x3_zeroed),
    PGAN_ENTRY(pgan_conv_h2_prep), PGAN_ENTRY(pgan_conv_h2_prep_multi), PGAN_ENTRY(pgan_conv_h2_forget),
    PGAN_ENTRY(pgan_conv_down_h2), PGAN_ENTRY(pgan_conv_up_h2), PGAN_ENTRY(pgan_conv_wgrad_h2), PGAN_ENTRY(pgan_conv_down_h2_zeroed),
    PGAN_ENTRY(pgan_gn_fwd), PGAN_ENTRY(pgan_gn_fwd_presummed), PGAN_ENTRY(pgan_gn_fwd_static), PGAN_ENTRY(pgan_gn_bwd),
    PGAN_ENTRY(pgan_gn_jvp), PGAN_ENTRY(pgan_gn_bwd2), PGAN_ENTRY(pgan_gn_bwd_onepass),
    PGAN_ENTRY(pgan_bn_fwd), PGAN_ENTRY(pgan_bn_bwd), PGAN_ENTRY(pgan_bn_fwd_sums), PGAN_ENTRY(pgan_bn_fwd_apply),
    PGAN_ENTRY(pgan_bn_bwd_sums), PGAN_ENTRY(pgan_bn_bwd_apply),
    PGAN_ENTRY(pgan_head_fwd), PGAN_ENTRY(pgan_head_seed), PGAN_ENTRY(pgan_head_wgrad),
    PGAN_ENTRY(pgan_interpolate), PGAN_ENTRY(pgan_gp_penalty), PGAN_ENTRY(pgan_div_penalty), PGAN_ENTRY(pgan_critic_loss),
    PGAN_ENTRY(pgan_sum_scale), PGAN_ENTRY(pgan_act_bwd), PGAN_ENTRY(pgan_affine), PGAN_ENTRY(pgan_zero_multi), PGAN_ENTRY(pgan_adam),
    PGAN_ENTRY(pgan_philox), PGAN_ENTRY(pgan_philox_slice), PGAN_ENTRY(pgan_take_batch), PGAN_ENTRY(pgan_take_batch_slice),
    PGAN_ENTRY(pgan_label_smooth), PGAN_ENTRY(pgan_mono_switch),
    PGAN_ENTRY(pgan_d1d_fwd), PGAN_ENTRY(pgan_d1d_bwd), PGAN_ENTRY(pgan_d1d_fwd_len), PGAN_ENTRY(pgan_d1d_bwd_len),
    PGAN_ENTRY(pgan_ct1d_up), PGAN_ENTRY(pgan_ct1d_down), PGAN_ENTRY(pgan_ct1d_wgrad), PGAN_ENTRY(pgan_gemm),
    PGAN_ENTRY(pgan_unpack_sites), PGAN_ENTRY(pgan_pack_sites),
    PGAN_ENTRY(pgan_fakebuf_step), PGAN_ENTRY(pgan_sample_genotypes),
    PGAN_ENTRY(pgan_event_record), PGAN_ENTRY(pgan_stream_wait_event), PGAN_ENTRY(pgan_stream_edge),
};
constexpr int NENTRIES = (int)(sizeof(ENTRIES) / sizeof(ENTRIES[0]));

struct TapeItem {
    int (*call)(const uint64_t*);
    int first, entry;       // index of the first argument word; index into ENTRIES
};
struct Tape {
    std::vector<TapeItem> items;
    std::vector<uint64_t> words;
};

}  // namespace

extern "C" {

// How the device code of this library was compiled, as far as it matters to a caller: kernels built WITH packed-fp32
// vector instructions returned wrong values beside other streams' kernels on this part, so the loader refuses a library
// that does not state "packed-fp32-ops=off" (the build defines PGAN_NO_PACKED_FP32 together with the compiler flag).
const char* pgan_build_info(void) {
#ifdef PGAN_NO_PACKED_FP32
    return "gfx950 packed-fp32-ops=off";
#else
    return "gfx950 packed-fp32-ops=on";
#endif
}

// ---- events and stream edges ------------------------------------------------------------------------------------
int pgan_event_create(void** ev) {
    if (!ev) return PGAN_EINVAL;
    hipEvent_t e = nullptr;
    if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) return PGAN_ELAUNCH;
    g_events_created.fetch_add(1);
    *ev = e;
    return PGAN_OK;
}

int pgan_event_destroy(void* ev) {
    if (!ev) return PGAN_EINVAL;
    return hipEventDestroy(static_cast<hipEvent_t>(ev)) == hipSuccess ? PGAN_OK : PGAN_ELAUNCH;
}

long long pgan_events_created(void) { return g_events_created.load(); }

int pgan_event_record(void* ev, hipStream_t st) {
    if (!ev) return PGAN_EINVAL;
    return hipEventRecord(static_cast<hipEvent_t>(ev), st) == hipSuccess ? PGAN_OK : PGAN_ELAUNCH;
}

int pgan_stream_wait_event(hipStream_t st, void* ev) {
    if (!ev) return PGAN_EINVAL;
    return hipStreamWaitEvent(st, static_cast<hipEvent_t>(ev), 0) == hipSuccess ? PGAN_OK : PGAN_ELAUNCH;
}

// work enqueued on `to` after this call waits for everything enqueued on `from` before it
int pgan_stream_edge(void* ev, hipStream_t from, hipStream_t to) {
    if (!ev) return PGAN_EINVAL;
    if (from == to) return PGAN_OK;
    if (hipEventRecord(static_cast<hipEvent_t>(ev), from) != hipSuccess) return PGAN_ELAUNCH;
    return hipStreamWaitEvent(to, static_cast<hipEvent_t>(ev), 0) == hipSuccess ? PGAN_OK : PGAN_ELAUNCH;
}

// ---- launch tape ------------------------------------------------------------------------------------------------
void* pgan_tape_create(void) { return new (std::nothrow) Tape(); }

int pgan_tape_destroy(void* tape) {
    if (!tape) return PGAN_EINVAL;
    delete static_cast<Tape*>(tape);
    return PGAN_OK;
}

int pgan_tape_arg_count(const char* entry_point) {
    if (!entry_point) return PGAN_EINVAL;
    for (int i = 0; i < NENTRIES; ++i)
        if (!std::strcmp(ENTRIES[i].name, entry_point)) return ENTRIES[i].nargs;
    return PGAN_EINVAL;
}

int pgan_tape_append(void* tape, const char* entry_point, const unsigned long long* words, int nwords) {
    if (!tape || !entry_point || (nwords > 0 && !words)) return PGAN_EINVAL;
    Tape* t = static_cast<Tape*>(tape);
    for (int i = 0; i < NENTRIES; ++i) {
        if (std::strcmp(ENTRIES[i].name, entry_point)) continue;
        if (ENTRIES[i].nargs != nwords) return PGAN_EINVAL;
        t->items.push_back(TapeItem{ENTRIES[i].call, (int)t->words.size(), i});
        for (int k = 0; k < nwords; ++k) t->words.push_back(words[k]);
        return PGAN_OK;
    }
    return PGAN_EINVAL;
}

int pgan_tape_length(void* tape) { return tape ? (int)static_cast<Tape*>(tape)->items.size() : PGAN_EINVAL; }

// Replays every noted call in order.  Returns 0, or stops at the first call that fails: *failed_at (may be NULL) gets
// its index and the call's own return code is returned.
int pgan_tape_replay(void* tape, int* failed_at) {
    if (!tape) return PGAN_EINVAL;
    const Tape* t = static_cast<const Tape*>(tape);
    const uint64_t* w = t->words.data();
    const int n = (int)t->items.size();
    for (int i = 0; i < n; ++i) {
        const int rc = t->items[i].call(w + t->items[i].first);
        if (rc != PGAN_OK) {
            if (failed_at) *failed_at = i;
            return rc;
        }
    }
    return PGAN_OK;
}

// name of the entry point noted at position `index` (for error messages); NULL outside the tape
const char* pgan_tape_entry_name(void* tape, int index) {
    if (!tape) return nullptr;
    const Tape* t = static_cast<const Tape*>(tape);
    if (index < 0 || index >= (int)t->items.size()) return nullptr;
    return ENTRIES[t->items[index].entry].name;
}

}  // extern "C"
