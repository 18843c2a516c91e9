// libevhip.so host side: handle, packed-weight table, workspace arena, row layouts, forward
// orchestration and the C ABI declared in include/evhip.h.
//
// Data layout in HBM: every activation is channels-last [rows][channels].
// Rows of a batch are laid out with zero gaps between utterances:
//     [G gap rows][utt 0 rows][G gap rows][utt 1 rows] ... [G gap rows][pad to a multiple of 256]
// G = 4 at token rate and at mel-frame rate; the vocoder's upsampled stages inherit the frame
// layout scaled by the cumulative upsampling factor (gap 32 / 256 / 512 / 1024 rows >= the largest
// conv halo of that stage, 25 rows).  Every kernel writes exact zeros into invalid rows, so a conv
// that reads across an utterance edge sees the zero padding the reference's per-utterance (B = 1)
// Conv1d(padding=...) provides, and no conv ever needs a per-row bounds test.
#include <hip/hip_runtime.h>

#include <math.h>
#include <stdarg.h>
#include <stdlib.h>
#include <stdio.h>
#include <string.h>

#include <algorithm>
#include <map>
#include <string>
#include <vector>

#include "../../include/evhip.h"
#include "../../include/evhip_ops.h"
#include "ev_kernels.h"

using namespace ev;

namespace {

constexpr int GAP = 4;            // gap rows between utterances (token and frame rate)
constexpr int ROW_ALIGN = 256;    // row counts are padded to the largest GEMM M tile
constexpr int PAD_ROWS = 64;      // readable slack rows before / after every activation buffer
constexpr int MEL_PAD = 96;       // n_mels padded to a multiple of 32 (MFMA K granularity)
constexpr size_t PIN_MAX_B = 1 << 16;                // utterances per call the pinned staging area is laid out for
constexpr size_t PIN_FRAME = 4 * PIN_MAX_B * 4;      // byte offset of the frame-layout region (after the token-layout region)
constexpr size_t PIN_BYTES = PIN_FRAME + 2 * PIN_MAX_B * 4;

thread_local std::string g_create_error;

struct WeightEntry { int dtype; int ndim; uint64_t dims[4]; const char* ptr; uint64_t nbytes; };

struct Buf {              // activation buffer with PAD_ROWS of slack on both sides
    char* base = nullptr; // allocation start
    char* p = nullptr;    // logical row 0
    size_t bytes = 0;
};

struct Tap { const void* ptr; int dtype; int ld; int C; int level; /* 0 token, 1 frame, 2+s vocoder stage s */ int shift; };

struct KStat { std::string name; int launches = 0; float ms = 0; double flops = 0, bytes = 0; };
struct PendingEvt { hipEvent_t a, b; int stat; int rec; };
struct LaunchRec { std::string name; int M = 0, N = 0, K = 0, taps = 0, dil = 0; float ms = 0; double flops = 0, bytes = 0; };

}  // namespace

struct ev_handle {
    ev_config cfg;
    int device = 0;
    hipStream_t own_stream = nullptr, stream = nullptr;
    hipStream_t aux[2] = {nullptr, nullptr};         // the first two ResBlocks of a generator stage run beside the third
    hipEvent_t ev_fork = nullptr, ev_join[2] = {nullptr, nullptr};
    std::string err;
    // weights
    char* wblob = nullptr; bool wblob_owned = false; size_t wbytes = 0;
    std::map<std::string, WeightEntry> wt;
    std::map<std::string, float> scalar_cache;
    float* pe_dev = nullptr; int pe_cap = 0;     // positional table, extended on demand beyond the packed length   // host copies of 1-element tensors (biases of the Linear(C,1) heads, PE alphas)
    // SimBERT style encoder (ev_style_load_weights / ev_style_embed): its own blob, merged into `wt` under the "sb." prefix
    char* sblob = nullptr; size_t sbytes = 0; ev_bert_config bcfg{}; bool style_loaded = false;
    // arena
    char* arena[3] = {nullptr, nullptr, nullptr}; size_t arena_bytes[3] = {0, 0, 0};   // [0] token-rate phase, [1] frame-rate phase + vocoder, [2] SimBERT
    char* tok_ks = nullptr; size_t tok_ks_bytes = 0;          // split-K partial sums of the token-rate conv-FFN (tok_splitk); inside arena 0
    char* pinned = nullptr; size_t pinned_bytes = 0;
    // persistent outputs (host side)
    std::vector<int32_t> mel_lens; std::vector<int64_t> mel_offs;
    std::vector<int64_t> forced_dur;
    std::vector<int64_t> pack_host[8]; std::vector<int32_t> pack_rows[8]; int pack_slot = 0;   // host staging of pack_level (kept alive, no sync)
    // layout of the last call
    int B = 0, total_tokens = 0; int64_t total_frames = 0;
    int Rt = 0, Rf = 0;
    std::vector<int32_t> tok_off, tok_len, frm_off;
    std::map<std::string, Tap> taps;
    const int64_t* last_dur = nullptr; const int32_t* last_mel_len_dev = nullptr;
    // device maps (inside the arena)
    int32_t *d_tok_seq = nullptr, *d_tok_pos = nullptr, *d_tok_off = nullptr, *d_tok_len = nullptr, *d_cu = nullptr;
    uint8_t* d_tok_valid = nullptr;
    int32_t *d_frm_seq = nullptr, *d_frm_pos = nullptr, *d_frm_off = nullptr, *d_mel_len = nullptr, *d_frm_len = nullptr;
    uint8_t* d_frm_valid = nullptr;
    // profiling
    bool profiling = false;
    std::vector<KStat> stats; std::map<std::string, int> stat_idx;
    std::vector<LaunchRec> launches;          // one record per launch of the last profiled call, in launch order
    std::vector<PendingEvt> pending; std::vector<hipEvent_t> evt_pool; size_t evt_next = 0;
    std::map<std::string, float> timings;
    std::map<std::string, std::pair<hipEvent_t, hipEvent_t>> region_evt;
};

namespace {

int fail(ev_handle* h, const char* fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (h) h->err = buf; else g_create_error = buf;
    return -1;
}

#define HIPCHK(h, expr)                                                                       \
    do {                                                                                      \
        hipError_t _e = (expr);                                                               \
        if (_e != hipSuccess) return fail(h, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
    } while (0)

size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }
int ilog2(int v) { int s = 0; while ((1 << s) < v) ++s; return s; }

// ---------------------------------------------------------------- weights
// Blob layout (emotivoice_amd/packer.py): "EVW1\0\0\0\0" | u32 count | u32 reserved |
// count x { char name[64]; u32 dtype(0 f16,1 f32,2 i32,3 i64); u32 ndim; u64 dims[4]; u64 offset; u64 nbytes } | data (256-B aligned)
struct BlobEntry { char name[64]; uint32_t dtype, ndim; uint64_t dims[4]; uint64_t offset, nbytes; };

// `style` selects which of the two blobs is being (re)loaded: the generator's (every name without the "sb." prefix) or the
// SimBERT encoder's ("sb." names); the other one's entries stay in the table.
int parse_blob(ev_handle* h, const char* host_hdr, size_t nbytes, bool style = false) {
    if (nbytes < 16 || memcmp(host_hdr, "EVW1", 4) != 0) return fail(h, "weight blob: bad magic");
    uint32_t count;
    memcpy(&count, host_hdr + 8, 4);
    if (16 + (size_t)count * sizeof(BlobEntry) > nbytes) return fail(h, "weight blob: truncated table");
    for (auto it = h->wt.begin(); it != h->wt.end();) {
        const bool is_style = it->first.compare(0, 3, "sb.") == 0;
        if (is_style == style) it = h->wt.erase(it); else ++it;
    }
    h->scalar_cache.clear();
    if (!style) {
        if (h->pe_dev) { (void)hipFree(h->pe_dev); h->pe_dev = nullptr; }
        h->pe_cap = 0;
    }
    char* base = style ? h->sblob : h->wblob;
    for (uint32_t i = 0; i < count; ++i) {
        BlobEntry e;
        memcpy(&e, host_hdr + 16 + (size_t)i * sizeof(BlobEntry), sizeof e);
        e.name[63] = 0;
        if (e.offset + e.nbytes > nbytes) return fail(h, "weight blob: tensor %s out of range", e.name);
        if ((strncmp(e.name, "sb.", 3) == 0) != style) return fail(h, "weight blob: tensor %s does not belong in the %s blob", e.name, style ? "SimBERT" : "generator");
        WeightEntry w;
        w.dtype = (int)e.dtype; w.ndim = (int)e.ndim; memcpy(w.dims, e.dims, sizeof w.dims);
        w.ptr = base + e.offset; w.nbytes = e.nbytes;
        h->wt[e.name] = w;
    }
    return 0;
}

const WeightEntry* W(ev_handle* h, const std::string& name) {
    auto it = h->wt.find(name);
    if (it == h->wt.end()) { h->err = "missing packed weight: " + name; return nullptr; }
    return &it->second;
}
#define WPTR(var, type, name)                                   \
    const type* var;                                            \
    {                                                           \
        const WeightEntry* _w = W(h, name);                     \
        if (!_w) return -1;                                     \
        var = reinterpret_cast<const type*>(_w->ptr);           \
    }

int get_scalar(ev_handle* h, const std::string& name, float* v) {
    auto it = h->scalar_cache.find(name);
    if (it == h->scalar_cache.end()) {
        const WeightEntry* w = W(h, name);
        if (!w) return -1;
        float x;
        HIPCHK(h, hipMemcpy(&x, w->ptr, 4, hipMemcpyDeviceToHost));
        it = h->scalar_cache.emplace(name, x).first;
    }
    *v = it->second;
    return 0;
}

// positional encoding rows [0, need): the packed (torch-exact) table first, longer utterances computed on the device
int ensure_pe(ev_handle* h, int need) {
    if (need <= h->pe_cap) return 0;
    const WeightEntry* pw = W(h, "pe");
    const WeightEntry* dw = W(h, "pe_div");
    if (!pw || !dw) return -1;
    const int C = h->cfg.hidden, packed = (int)pw->dims[0];
    int cap = std::max(need, std::max(packed, 2 * h->pe_cap));
    cap = (int)align_up((size_t)cap, 1024);
    float* nb = nullptr;
    HIPCHK(h, hipMalloc((void**)&nb, (size_t)cap * C * 4));
    const int have = std::min(packed, cap);
    HIPCHK(h, hipMemcpyAsync(nb, pw->ptr, (size_t)have * C * 4, hipMemcpyDeviceToDevice, h->stream));
    launch_pe_extend(nb, reinterpret_cast<const float*>(dw->ptr), have, cap, C, h->stream);
    HIPCHK(h, hipStreamSynchronize(h->stream));
    if (h->pe_dev) HIPCHK(h, hipFree(h->pe_dev));
    h->pe_dev = nb; h->pe_cap = cap;
    return 0;
}

// ---------------------------------------------------------------- arena
int arena_reserve(ev_handle* h, int idx, size_t bytes) {
    if (bytes <= h->arena_bytes[idx]) return 0;
    if (h->arena[idx]) { HIPCHK(h, hipStreamSynchronize(h->stream)); HIPCHK(h, hipFree(h->arena[idx])); h->arena[idx] = nullptr; h->arena_bytes[idx] = 0; }
    bytes = align_up(bytes + bytes / 8, 1 << 20);
    HIPCHK(h, hipMalloc((void**)&h->arena[idx], bytes));
    HIPCHK(h, hipMemsetAsync(h->arena[idx], 0, bytes, h->stream));
    h->arena_bytes[idx] = bytes;
    return 0;
}
struct ArenaPlan {   // two-pass: the dry pass measures, the second pass hands out pointers
    ev_handle* h; int idx; bool dry; size_t off = 0;
    char* take(size_t bytes) {
        off = align_up(off, 256);
        char* p = dry ? nullptr : h->arena[idx] + off;
        off += bytes;
        return p;
    }
    Buf rows(size_t rows, size_t ld, size_t es) {
        Buf b;
        const size_t pad = (size_t)PAD_ROWS * ld * es;
        b.bytes = rows * ld * es;
        b.base = take(pad + b.bytes + pad);
        b.p = dry ? nullptr : b.base + pad;
        return b;
    }
    template <typename T> T* arr(size_t n) { return reinterpret_cast<T*>(take(n * sizeof(T))); }
};

int pinned_reserve(ev_handle* h, size_t bytes) {
    if (bytes <= h->pinned_bytes) return 0;
    if (h->pinned) { HIPCHK(h, hipStreamSynchronize(h->stream)); HIPCHK(h, hipHostFree(h->pinned)); h->pinned = nullptr; }
    bytes = align_up(bytes * 2, 1 << 16);
    HIPCHK(h, hipHostMalloc((void**)&h->pinned, bytes, hipHostMallocDefault));
    h->pinned_bytes = bytes;
    return 0;
}

// ---------------------------------------------------------------- profiling helpers
int stat_id(ev_handle* h, const char* name) {
    auto it = h->stat_idx.find(name);
    if (it != h->stat_idx.end()) return it->second;
    KStat s; s.name = name;
    h->stats.push_back(s);
    h->stat_idx[name] = (int)h->stats.size() - 1;
    return (int)h->stats.size() - 1;
}
hipEvent_t get_evt(ev_handle* h) {
    if (h->evt_next == h->evt_pool.size()) { hipEvent_t e; (void)hipEventCreate(&e); h->evt_pool.push_back(e); }
    return h->evt_pool[h->evt_next++];
}
struct KScope {   // wraps one kernel launch with events when profiling is on
    ev_handle* h; int sid = -1; int rec = -1; hipEvent_t a{}, b{}; hipStream_t st;
    KScope(ev_handle* h_, const char* name, double flops, double bytes, hipStream_t s = nullptr, const ConvGemmParams* g = nullptr)
        : h(h_), st(s ? s : h_->stream) {
        if (!h->profiling) return;
        sid = stat_id(h, name);
        h->stats[sid].launches++; h->stats[sid].flops += flops; h->stats[sid].bytes += bytes;
        LaunchRec r; r.name = name; r.flops = flops; r.bytes = bytes;
        if (g) { r.M = g->M; r.N = g->N; r.K = g->K; r.taps = g->taps; r.dil = g->dil; }
        h->launches.push_back(r);
        rec = (int)h->launches.size() - 1;
        a = get_evt(h); b = get_evt(h);
        (void)hipEventRecord(a, st);
    }
    ~KScope() {
        if (sid < 0) return;
        (void)hipEventRecord(b, st);
        h->pending.push_back({a, b, sid, rec});
    }
};
void region_begin(ev_handle* h, const char* name) {
    if (!h->profiling) return;
    auto& pr = h->region_evt[name];
    if (!pr.first) { (void)hipEventCreate(&pr.first); (void)hipEventCreate(&pr.second); }
    (void)hipEventRecord(pr.first, h->stream);
}
void region_end(ev_handle* h, const char* name) {
    if (!h->profiling) return;
    (void)hipEventRecord(h->region_evt[name].second, h->stream);
}
void profiling_reset(ev_handle* h) {
    h->stats.clear(); h->stat_idx.clear(); h->pending.clear(); h->evt_next = 0; h->timings.clear(); h->launches.clear();
}
void profiling_collect(ev_handle* h) {
    if (!h->profiling) return;
    for (auto& p : h->pending) {
        float ms = 0;
        (void)hipEventElapsedTime(&ms, p.a, p.b);
        h->stats[p.stat].ms += ms;
        if (p.rec >= 0 && p.rec < (int)h->launches.size()) h->launches[p.rec].ms = ms;
    }
    h->pending.clear();
    for (auto& kv : h->region_evt) {
        float ms = 0;
        if (hipEventElapsedTime(&ms, kv.second.first, kv.second.second) == hipSuccess) h->timings[kv.first] = ms;
    }
}

// ---------------------------------------------------------------- launch wrappers with precondition checks
int check_gemm(ev_handle* h, const ConvGemmParams& p) {
    const int es = p.dtype == DT_F16 ? 2 : 4;
    if (p.M % ROW_ALIGN) return fail(h, "gemm: M=%d not a multiple of %d", p.M, ROW_ALIGN);
    if (p.dtype == DT_F32S && (p.K % 32 || !p.W_lo)) return fail(h, "gemm: bad split-precision call");
    if (p.pro_lrelu && !(p.pro_slope >= 0.f && p.pro_slope <= 1.f)) return fail(h, "gemm: prologue leaky-relu slope %g outside [0, 1]", p.pro_slope);
    if (p.N % 32) return fail(h, "gemm: N=%d not a multiple of 32", p.N);
    if ((p.K * es) % 64) return fail(h, "gemm: K=%d not a multiple of %d", p.K, 64 / es);
    if ((p.taps - 1) * p.dil > 64) return fail(h, "gemm: conv span %d > 64", (p.taps - 1) * p.dil);
    if (p.center * p.dil > PAD_ROWS || (p.taps - 1 - p.center) * p.dil > PAD_ROWS) return fail(h, "gemm: halo exceeds buffer slack");
    if ((p.lda * es) % 16 || ((uintptr_t)p.A & 15) || ((uintptr_t)p.W & 15)) return fail(h, "gemm: unaligned operand");
    if (!p.out16 && !p.out32 && !p.mxo_h) return fail(h, "gemm: no output");
    if (p.dtype == DT_MX && (p.K % 32 || !p.W)) return fail(h, "gemm: bad MX call");
    if (mx_check(p)) return fail(h, "gemm: inconsistent MX plane-set fields (dtype %d, N %d, K %d, taps %d)", p.dtype, p.N, p.K, p.taps);
    if (splitk_check(p)) return fail(h, "gemm: inconsistent split-K call (ksplit %d, dtype %d, N %d, K %d)", p.ksplit, p.dtype, p.N, p.K);
    return 0;
}
// flop_scale: algorithmic / executed FLOPs (2/3 for a ConvTranspose1d run as a 3-tap conv: each output sample has two real taps)
int gemm(ev_handle* h, const char* name, const ConvGemmParams& p, double valid_rows, hipStream_t st = nullptr, double flop_scale = 1.0) {
    if (check_gemm(h, p)) return -1;
    const int es = p.dtype == DT_F16 ? 2 : 4;
    const double flops = 2.0 * valid_rows * p.N * (double)p.K * p.taps * flop_scale;
    double bytes = valid_rows * ((double)p.K * es + (double)p.N * (p.out16 ? 2 : 0) + (double)p.N * (p.out32 ? 4 : 0)) +
                   (double)p.N * p.K * p.taps * es;
    // a DT_MX call that launch_conv_gemm will run as the split-precision kernel (no plane-set input, unsupported shape) is recorded as such
    if (p.dtype == DT_MX && mx_launch_kind(p) == 0) name = strncmp(name, "dec", 3) == 0 ? "dec_f32_gemm" : "voc_conv_gemm_x3";
    KScope ks(h, name, flops, bytes, st, &p);
    launch_conv_gemm(p, st ? st : h->stream);
    return 0;
}
// three independent convs of one level of a stage's ResBlocks as one grid (launch_conv_gemm_group3), or one by one when they are not such a triple
int gemm_group3(ev_handle* h, const char* name, const ConvGemmParams* ps, double valid_rows) {
    double flops = 0, bytes = 0;
    int taps = 0;
    for (int i = 0; i < 3; ++i) {
        if (check_gemm(h, ps[i])) return -1;
        flops += 2.0 * valid_rows * ps[i].N * (double)ps[i].K * ps[i].taps;
        bytes += valid_rows * ((double)ps[i].K * 4 + (double)ps[i].N * (ps[i].out32 ? 4 : 0)) + (double)ps[i].N * ps[i].K * ps[i].taps * 4;
        taps += ps[i].taps;
    }
    if (launch_conv_gemm_group3(ps, h->stream, true) == 0) {
        ConvGemmParams shape = ps[0];
        shape.taps = taps; shape.dil = 0;          // (the record of a grouped launch: the three convs' taps summed, no single dilation)
        KScope ks(h, name, flops, bytes, nullptr, &shape);
        return launch_conv_gemm_group3(ps, h->stream);
    }
    for (int i = 0; i < 3; ++i)          // not a triple the grouped kernel takes: one by one
        if (gemm(h, name, ps[i], valid_rows)) return -1;
    return 0;
}
ConvGemmParams gemm_defaults() {
    ConvGemmParams p;
    memset(&p, 0, sizeof p);
    p.taps = 1; p.dil = 1; p.center = 0; p.out_scale = 1.0f;
    return p;
}

// token-rate (fp32) GEMM operands: exact fp32 MFMA or the hi/lo split pair (cfg.token_rate_split)
int tok_weights(ev_handle* h, const std::string& base /* e.g. "enc.0.qkv.w" */, ConvGemmParams& p) {
    const bool dec = base.compare(0, 4, "dec.") == 0 || base.compare(0, 7, "to_mel.") == 0;
    if (dec ? (h->cfg.decoder_precision == EV_PREC_X3 || h->cfg.decoder_precision == EV_PREC_MX) : h->cfg.token_rate_split != 0) {
        // hi part: fp16(w).  The token-rate stack packs it as "<name>32h"; for the decoder it is the fp16 copy "<name>16"
        const WeightEntry* hi = W(h, base + (h->wt.count(base + "32h") ? "32h" : "16"));
        const WeightEntry* lo = W(h, base + "32l");
        if (!hi || !lo) return -1;
        p.dtype = DT_F32S; p.W = hi->ptr; p.W_lo = lo->ptr;
    } else {
        const WeightEntry* w = W(h, base + "32");
        if (!w) return -1;
        p.dtype = DT_F32; p.W = w->ptr; p.W_lo = nullptr;
    }
    return 0;
}

// raw storage of one MX plane set (ev_gemm_mx.h), sized for the largest [rows][C] it will hold; mx_view() lays a tensor into it
struct PlaneBuf { char* h = nullptr; char* q4[2] = {nullptr, nullptr}; char* qs[2] = {nullptr, nullptr}; };
struct MxView { char* h; char* q4[2]; char* qs[2]; unsigned qs_stride; int logC; };
static constexpr size_t MX_PAD = 64;        // slack rows of every plane, both sides
MxView mx_view(const PlaneBuf& b, size_t rows, int C) {
    MxView v;
    v.h = b.h + MX_PAD * C * 2;
    for (int i = 0; i < 2; ++i) { v.q4[i] = b.q4[i] + MX_PAD * (C / 2); v.qs[i] = b.qs[i] + MX_PAD * 4; }
    v.qs_stride = (unsigned)((rows + 2 * MX_PAD) * 4);
    v.logC = ilog2(C);
    return v;
}
void mx_out(ConvGemmParams& p, const MxView& v, float slope) {
    p.mxo_h = v.h; p.mxo_q4[0] = v.q4[0]; p.mxo_q4[1] = v.q4[1]; p.mxo_qs[0] = v.qs[0]; p.mxo_qs[1] = v.qs[1];
    p.mxo_qs_stride = v.qs_stride; p.mxo_logC = v.logC; p.mxo_slope = slope;
}
void mx_in(ConvGemmParams& p, const MxView& v, int C) {
    p.A = v.h; p.lda = C; p.pro_lrelu = 0;
    p.mx_x4[0] = v.q4[0]; p.mx_x4[1] = v.q4[1]; p.mx_xs[0] = v.qs[0]; p.mx_xs[1] = v.qs[1]; p.mx_xs_stride = v.qs_stride;
}

// EV_PREC_MX decoder: plane set of the conv-FFN's hidden activation [R][4C] and the planes-kernel scratch of its fp32 input [R][C]
struct DecMx { PlaneBuf ffn; char* scratch; size_t scratch_bytes; };

struct RowCtx {     // one row layout (token rate or frame rate)
    int R; const uint8_t* valid; const int32_t* row_seq; const int32_t* seq_off; const int32_t* seq_len; int B; int max_len;
    double n_valid;
};

// Split-K of a token-rate GEMM (ev_config.token_splitk; ConvGemmParams::ksplit): a rule on the layer's shape only -- never on the row count -- so that an
// utterance alone and inside a batch gets the same summation order.  It takes the conv-FFN's second conv (N = hidden, 144 (K-chunk, tap) steps: 4 ranges of 36);
// N <= hidden keeps the partial-sum traffic (ksplit x M x N x 4 bytes, written and read once) small beside the operands.  Measured (bench.py --token-splitk,
// one MI355X): B = 1, 64 / 256 phonemes 4.25 / 5.2 -> 3.95 / 4.85 ms; at 32 x 256 tokens the encoder pays +0.08 ms for the reduction launches.  The predictors'
// k = 3 convs (36 steps) were tried with 3 ranges as well: another -0.07 ms at B = 1 for +0.14 ms at B = 32 -- left in one pass.
void tok_splitk(ev_handle* h, ConvGemmParams& p) {
    if (h->cfg.token_splitk != 0 || p.dtype != DT_F32S || p.N % 64 || p.N > h->cfg.hidden || p.add16_a) return;
    const int nkc = p.K / 32, steps = nkc * p.taps;
    const int S = (steps >= 144 && nkc % 4 == 0) ? 4 : 1;
    if (S <= 1 || !h->tok_ks || (size_t)S * (size_t)p.M * (size_t)p.N * 4 > h->tok_ks_bytes) return;          // (the arena sizes the buffer for the rule's worst case)
    p.ksplit = S; p.mx_scratch = h->tok_ks; p.mx_scratch_size = h->tok_ks_bytes;
}

// Encoder / decoder stack (reference modules/encoder.py:316-324, layer :154-200).  x (fp32 residual stream,
// [R][C]) is updated in place; y receives after_norm(x) in `prec` dtype (and y32_tap in fp32 if given).
int run_stack(ev_handle* h, const char* pre, int layers, int prec, const RowCtx& rc, Buf& x, Buf& hbuf, Buf& qkv, Buf& ctx, Buf& ffn,
              Buf& y, float* y32_tap, std::vector<Buf>* layer_taps, const DecMx* dmx = nullptr) {
    const int C = h->cfg.hidden, F = 4 * C, kf = h->cfg.ffn_kernel;
    const char* wsuf = prec == DT_F16 ? "w16" : "w32";
    const std::string sp(pre);
    const std::string kn = std::string(pre) + (prec == DT_F16 ? "_f16" : "_f32");
    for (int i = 0; i < layers; ++i) {
        const std::string lp = sp + "." + std::to_string(i);
        WPTR(g1, float, lp + ".ln1.g"); WPTR(b1, float, lp + ".ln1.b");
        WPTR(g2, float, lp + ".ln2.g"); WPTR(b2, float, lp + ".ln2.b");
        WPTR(wqkv, char, lp + ".qkv." + wsuf); WPTR(bqkv, float, lp + ".qkv.b");
        WPTR(wout, char, lp + ".out." + wsuf); WPTR(bout, float, lp + ".out.b");
        WPTR(wf1, char, lp + ".ffn1." + wsuf); WPTR(bf1, float, lp + ".ffn1.b");
        WPTR(wf2, char, lp + ".ffn2." + wsuf); WPTR(bf2, float, lp + ".ffn2.b");
        ConvGemmParams p = gemm_defaults();
        p.dtype = prec; p.A = hbuf.p; p.lda = C; p.W = wqkv; p.bias = bqkv; p.M = rc.R; p.N = 3 * C; p.K = C;
        p.row_valid = rc.valid; p.ldo = 3 * C;
        if (prec == DT_F32 && tok_weights(h, lp + ".qkv.w", p)) return -1;
        if (prec == DT_F16) p.out16 = qkv.p; else p.out32 = (float*)qkv.p;
        // EV_PREC_MX decoder: the QKV / output projections on the one-tap MX GEMM (ev_gemm_mx1.h); their fp32 inputs become plane sets in the scratch
        const bool lin_mx = dmx && p.dtype == DT_F32S && h->wt.count(lp + ".qkv.wmx") && h->wt.count(lp + ".out.wmx");
        // ... and the two LayerNorms write the plane sets their consumers (QKV, the conv-FFN's first conv) read straight into that scratch: no fp32 copy of
        // LN(x) and no mx_planes_kernel pass over it (ev_config.decoder_ln_planes = 1 restores the two passes; the planes are the same bits either way)
        const bool ln_pl = dmx && h->cfg.decoder_ln_planes == 0 && C <= 512 && dmx->scratch_bytes >= mx_scratch_bytes(rc.R, C);
        const MxScratchPlanes lnp = ln_pl ? mx_scratch_planes(dmx->scratch, rc.R, C) : MxScratchPlanes{};
        auto ln_to_planes = [&](LayerNormParams& l) {
            l.out32 = nullptr; l.out16 = nullptr;
            l.mxo_h = lnp.h; l.mxo_q4[0] = lnp.q4[0]; l.mxo_q4[1] = lnp.q4[1]; l.mxo_qs[0] = lnp.qs[0]; l.mxo_qs[1] = lnp.qs[1]; l.mxo_qs_stride = lnp.qs_stride;
        };
        auto planes_in = [&](ConvGemmParams& q) {
            q.A = lnp.h; q.lda = C; q.mx_x4[0] = lnp.q4[0]; q.mx_x4[1] = lnp.q4[1]; q.mx_xs[0] = lnp.qs[0]; q.mx_xs[1] = lnp.qs[1]; q.mx_xs_stride = lnp.qs_stride;
        };
        LayerNormParams ln{};
        ln.x = (const float*)x.p; ln.ldx = C; ln.rows = rc.R; ln.C = C; ln.gamma = g1; ln.beta = b1; ln.eps = 1e-12f;
        ln.row_valid = rc.valid; ln.ldo = C;
        if (prec == DT_F16) ln.out16 = hbuf.p; else ln.out32 = (float*)hbuf.p;
        if (lin_mx && ln_pl) ln_to_planes(ln);
        { KScope ks(h, "layernorm", 0, rc.n_valid * C * 6.0); launch_layernorm(ln, h->stream); }
        if (lin_mx) { p.dtype = DT_MX; p.W_mx = h->wt[lp + ".qkv.wmx"].ptr; p.mx_scratch = dmx->scratch; p.mx_scratch_size = dmx->scratch_bytes; }
        if (lin_mx && ln_pl) planes_in(p);
        if (gemm(h, lin_mx ? (std::string(pre) + "_mx_gemm").c_str() : (kn + "_gemm").c_str(), p, rc.n_valid)) return -1;
        AttnParams ap{};
        // decoder in the strict / mx modes: split-precision attention (three fp16 MFMAs per product); the token-rate encoder keeps exact fp32
        const bool att_split = prec == DT_F32 && !strcmp(pre, "dec") && (h->cfg.decoder_precision == EV_PREC_X3 || h->cfg.decoder_precision == EV_PREC_MX) &&
                               C / h->cfg.heads == 48 && h->cfg.decoder_attention == 0;
        ap.qkv = qkv.p; ap.dtype = att_split ? (int)DT_F32S : prec; ap.ld = 3 * C; ap.C = C; ap.heads = h->cfg.heads; ap.seq_off = rc.seq_off;
        ap.seq_len = rc.seq_len; ap.B = rc.B; ap.max_len = rc.max_len; ap.out = ctx.p; ap.ldo = C;
        { KScope ks(h, (kn + "_attention").c_str(), 0, 0); launch_attention(ap, h->stream); }
        p = gemm_defaults();
        p.dtype = prec; p.A = ctx.p; p.lda = C; p.W = wout; p.bias = bout; p.M = rc.R; p.N = C; p.K = C;
        p.row_valid = rc.valid; p.res = x.p; p.res_dtype = DT_F32; p.ldres = C; p.out32 = (float*)x.p; p.ldo = C;
        if (prec == DT_F32 && tok_weights(h, lp + ".out.w", p)) return -1;
        if (lin_mx) { p.dtype = DT_MX; p.W_mx = h->wt[lp + ".out.wmx"].ptr; p.mx_scratch = dmx->scratch; p.mx_scratch_size = dmx->scratch_bytes; }
        if (gemm(h, lin_mx ? (std::string(pre) + "_mx_gemm").c_str() : (kn + "_gemm").c_str(), p, rc.n_valid)) return -1;
        p = gemm_defaults();
        p.dtype = prec; p.A = hbuf.p; p.lda = C; p.W = wf1; p.bias = bf1; p.M = rc.R; p.N = F; p.K = C; p.taps = kf; p.center = (kf - 1) / 2;
        p.row_valid = rc.valid; p.act = ACT_GELU; p.ldo = F;
        if (prec == DT_F32 && tok_weights(h, lp + ".ffn1.w", p)) return -1;
        // EV_PREC_MX decoder: the conv-FFN (72 % of the stack's FLOPs) on the MX kernel; its hidden activation only exists as conv2's operand planes
        const bool ffn_mx = dmx && p.dtype == DT_F32S && h->wt.count(lp + ".ffn1.wmx") && h->wt.count(lp + ".ffn2.wmx");
        ln.gamma = g2; ln.beta = b2;
        if (prec == DT_F16) ln.out16 = hbuf.p; else ln.out32 = (float*)hbuf.p;
        ln.mxo_h = nullptr;
        if (ffn_mx && ln_pl) ln_to_planes(ln);
        { KScope ks(h, "layernorm", 0, rc.n_valid * C * 6.0); launch_layernorm(ln, h->stream); }
        MxView fv{};
        if (ffn_mx) {
            fv = mx_view(dmx->ffn, (size_t)rc.R, F);
            fv.logC = 0;                                          // dense [R][F] geometry (F = 1536 is not a power of two)
            p.dtype = DT_MX; p.W_mx = h->wt[lp + ".ffn1.wmx"].ptr; p.mx_scratch = dmx->scratch; p.mx_scratch_size = dmx->scratch_bytes;
            if (ln_pl) planes_in(p);
            mx_out(p, fv, 1.0f);
        } else if (prec == DT_F16) p.out16 = ffn.p; else p.out32 = (float*)ffn.p;
        if (gemm(h, ffn_mx ? (std::string(pre) + "_mx_gemm").c_str() : (kn + "_gemm").c_str(), p, rc.n_valid)) return -1;
        p = gemm_defaults();
        p.dtype = prec; p.A = ffn.p; p.lda = F; p.W = wf2; p.bias = bf2; p.M = rc.R; p.N = C; p.K = F; p.taps = kf; p.center = (kf - 1) / 2;
        p.row_valid = rc.valid; p.res = x.p; p.res_dtype = DT_F32; p.ldres = C; p.out32 = (float*)x.p; p.ldo = C;
        if (prec == DT_F32 && tok_weights(h, lp + ".ffn2.w", p)) return -1;
        if (!dmx && !strcmp(pre, "enc")) tok_splitk(h, p);
        if (ffn_mx) { p.dtype = DT_MX; p.W_mx = h->wt[lp + ".ffn2.wmx"].ptr; mx_in(p, fv, F); }
        if (gemm(h, ffn_mx ? (std::string(pre) + "_mx_gemm").c_str() : (kn + "_gemm").c_str(), p, rc.n_valid)) return -1;
        if (layer_taps) HIPCHK(h, hipMemcpyAsync((*layer_taps)[i].p, x.p, x.bytes, hipMemcpyDeviceToDevice, h->stream));
    }
    WPTR(ga, float, sp + ".after.g"); WPTR(ba, float, sp + ".after.b");
    LayerNormParams ln{};
    ln.x = (const float*)x.p; ln.ldx = C; ln.rows = rc.R; ln.C = C; ln.gamma = ga; ln.beta = ba; ln.eps = 1e-12f;
    ln.row_valid = rc.valid; ln.ldo = C;
    if (prec == DT_F16) { ln.out16 = y.p; ln.out32 = y32_tap; } else { ln.out32 = (float*)y.p; }
    { KScope ks(h, "layernorm", 0, rc.n_valid * C * 6.0); launch_layernorm(ln, h->stream); }
    return 0;
}

// Variance / duration predictor (reference modules/variance.py:36-56, 101-124): n x [conv k3 -> ReLU -> LN] -> Linear(C,1)
int run_predictor(ev_handle* h, const char* name, int layers, const RowCtx& rc, const Buf& xin, Buf& t1, Buf& t2, float* out_rows,
                  hipStream_t st = nullptr) {
    if (!st) st = h->stream;
    const int C = h->cfg.hidden;
    const std::string sp(name);
    const void* cur = xin.p;
    for (int i = 0; i < layers; ++i) {
        const std::string lp = sp + "." + std::to_string(i);
        WPTR(w, char, lp + ".conv.w32"); WPTR(b, float, lp + ".conv.b");
        // kernel size = the packed weight's tap dimension ([C][k][C]): duration / pitch / energy predictors have their own sizes in
        // the reference (model_open_source.py:46-76: duration_kernel_size, variance_kernel_size, energy hard-coded to 3)
        const int k = (int)W(h, lp + ".conv.w32")->dims[1];
        if ((k - 1) / 2 > GAP) return fail(h, "%s: kernel %d needs a halo of %d rows > the %d gap rows between utterances", lp.c_str(), k, (k - 1) / 2, GAP);
        WPTR(g, float, lp + ".ln.g"); WPTR(be, float, lp + ".ln.b");
        ConvGemmParams p = gemm_defaults();
        p.dtype = DT_F32; p.A = cur; p.lda = C; p.W = w; p.bias = b; p.M = rc.R; p.N = C; p.K = C; p.taps = k; p.center = (k - 1) / 2;
        p.row_valid = rc.valid; p.act = ACT_RELU; p.out32 = (float*)t1.p; p.ldo = C;
        if (tok_weights(h, lp + ".conv.w", p)) return -1;
        if (gemm(h, "variance_f32_gemm", p, rc.n_valid, st)) return -1;
        LayerNormParams ln{};
        ln.x = (const float*)t1.p; ln.ldx = C; ln.rows = rc.R; ln.C = C; ln.gamma = g; ln.beta = be; ln.eps = 1e-12f;
        ln.row_valid = rc.valid; ln.ldo = C;
        if (i + 1 < layers) {
            ln.out32 = (float*)t2.p;
        } else {
            WPTR(lw, float, sp + ".lin.w");
            float lbv;
            if (get_scalar(h, sp + ".lin.b", &lbv)) return -1;
            ln.dot_w = lw; ln.dot_b = lbv; ln.dot_out = out_rows;
        }
        { KScope ks(h, "layernorm", 0, rc.n_valid * C * 8.0, st); launch_layernorm(ln, st); }
        cur = t2.p;
    }
    return 0;
}

struct VocBufs {
    Buf pre, xu[4], tmp[3], rba[3], rbb[3], nxt[4], mrf32, mrf16a, mrf16b, wavrows; Buf mrf_tap[4]; Buf pre_tap;
    // EV_PREC_MX: plane sets of the up-conv output, conv1's output, the two alternating ResBlock states and the stage output; the
    // planes-kernel scratch of the one fp32 tensor an MX launch reads (conv_pre's output)
    // (pl_t / pl_a / pl_b: one set per ResBlock of a stage when the three run concurrently -- small batches, voc_small_batch -- else only [0])
    // pl_mrf: the running MRF sum of a stage as a PARTIAL plane set (hi plane, remainder codes, their scales; ev_config.mx_mrf == 0)
    PlaneBuf pl_xu, pl_t[3], pl_a[3], pl_b[3], pl_nxt, pl_mrf; char* mx_scratch = nullptr; size_t mx_scratch_bytes = 0;
};

// Small batches (single utterances: the reference's own call pattern): a generator launch is a handful of tiles whose K loops are sequential chains
// (tools/probe_b1.py: 39 conv launches of ~35 us each for 64 phonemes while 240 of 256 CUs idle), so the three ResBlocks of a stage -- independent until
// their scaled outputs meet in the MRF sum -- run on three streams there, also in the mx mode (each with its own intermediates; the running fp32 sum
// keeps its order rb0, rb1, rb2 through events, so the result has the same bits as the serial order).  At full batches a launch fills the chip and
// streams only reorder the same work (measured in round 3: 54.99 vs 54.56 ms), so large batches stay serial and allocate one set of intermediates.
static bool voc_small_batch(int Rf) { return Rf <= 2048; }

// weights of one generator conv: fp16 (also the "hi" part of the split) and, in the split-precision mode, the "lo" part
int voc_weights(ev_handle* h, const std::string& base /* e.g. "voc.rb3.c1.0" */, bool x3, ConvGemmParams& p, bool mx = false) {
    const WeightEntry* w = W(h, base + ".w16");
    const WeightEntry* b = W(h, base + ".b");
    if (!w || !b) return -1;
    p.W = w->ptr; p.bias = reinterpret_cast<const float*>(b->ptr);
    if (x3) {
        const WeightEntry* lo = W(h, base + ".w16l");
        if (!lo) return -1;
        p.dtype = DT_F32S; p.W_lo = lo->ptr;
        if (mx) {           // layers with fp4 planes in the blob (N, K % 128 == 0, 3 / 7 / 11 taps: packer.py) take the MX kernel
            auto it = h->wt.find(base + ".wmx");
            if (it == h->wt.end()) it = h->wt.find(base + ".wcmx");          // C = 64 layers: conv_c64_mx_kernel's planes (fp32 in / out)
            if (it != h->wt.end()) { p.dtype = DT_MX; p.W_mx = it->second.ptr; }
        }
    } else {
        p.dtype = DT_F16; p.W_lo = nullptr;
    }
    return 0;
}

// HiFi-GAN generator (reference models/hifigan/models.py:115-131) on channels-last rows.
// vocoder_precision F16: fp16 activations, every leaky-relu fused into the producer (post_lrelu) or the consumer's staging.
// vocoder_precision X3:  fp32 activations, split-precision products; every stored tensor is the RAW module output of the
// reference (so the Appendix-C taps are the buffers themselves) and each consumer applies its leaky-relu while staging.
int run_vocoder(ev_handle* h, const Buf& melin, int Rf, double n_frames, VocBufs& vb, bool keep) {
    const ev_config& c = h->cfg;
    // EV_PREC_MX: the split-precision data flow (fp32 raw module outputs), but every layer with N, K % 128 == 0 evaluates its products as
    // one fp16 MFMA + two block-scaled fp4 MFMAs (conv_gemm_mx_kernel) and reads its operand as the plane set its producer's epilogue wrote.
    const bool mx = c.vocoder_precision == EV_PREC_MX;
    const bool x3 = c.vocoder_precision == EV_PREC_X3 || mx;
    const size_t ves = x3 ? 4 : 2;
    auto set_out = [&](ConvGemmParams& q, void* dst) { if (x3) q.out32 = (float*)dst; else q.out16 = dst; };
    auto has_wt = [&](const std::string& name) { return h->wt.find(name) != h->wt.end(); };
    auto has_mx = [&](const std::string& base) { return mx && (has_wt(base + ".wmx") || has_wt(base + ".wcmx")); };
    bool prev_planes = false;           // vb.pl_nxt holds the plane set of lrelu(prev)
    ConvGemmParams p = gemm_defaults();
    if (voc_weights(h, "voc.pre", x3, p)) return -1;
    p.A = melin.p; p.lda = MEL_PAD; p.M = Rf; p.N = c.up_init_ch; p.K = MEL_PAD;
    p.taps = 7; p.center = 3; p.row_valid = h->d_frm_valid; p.valid_shift = 0; p.ldo = c.up_init_ch;
    set_out(p, vb.pre.p);
    if (!x3) {
        p.post_lrelu = 1; p.post_slope = 0.1f;      // leaky_relu(0.1) of models.py:118 fused into the producer
        if (keep) { p.out32 = (float*)vb.pre_tap.p; p.out32_before_post = 1; }
    }
    if (gemm(h, x3 ? "voc_conv_gemm_x3" : "voc_conv_gemm_f16", p, n_frames)) return -1;
    const char* gname = x3 ? "voc_conv_gemm_x3" : "voc_conv_gemm_f16";
    const void* prev = vb.pre.p;
    int ch = c.up_init_ch, U = 1;
    for (int i = 0; i < c.n_up; ++i) {
        const int s = c.up_rates[i], cout = ch / 2;
        const int rows_in = Rf * U, rows_out = rows_in * s;
        const double valid_in = n_frames * U, valid_out = valid_in * s;
        // ConvTranspose1d(k = 2s, pad = s/2) == 3-tap conv with N = s * C_out, viewed as [rows_in*s][C_out] (models.py:119)
        p = gemm_defaults();
        if (voc_weights(h, "voc.up" + std::to_string(i), x3, p, mx)) return -1;
        p.A = prev; p.lda = ch; p.M = rows_in; p.N = s * cout; p.K = ch; p.taps = 3; p.center = 1;
        if (cout % 64 == 0 && s % 2 == 0) p.polyphase_cout = cout;          // (conv_gemm_mx_kernel skips the zero tap of each output phase; other kernels ignore the hint)
        p.row_valid = h->d_frm_valid; p.valid_shift = ilog2(U); p.ldo = s * cout;
        if (x3) { p.pro_lrelu = 1; p.pro_slope = 0.1f; }          // models.py:118 (the fp16 path has it in the producer's epilogue)
        set_out(p, vb.xu[i].p);
        // the ResBlocks of this stage run on the MX kernel iff all their convs have fp4 planes (shape rule of the packer)
        bool stage_mx = mx && cout % 64 == 0;            // (C = 64: conv_c64_mx_kernel, the same plane-set data flow)
        for (int j = 0; stage_mx && j < c.n_rb; ++j) stage_mx = has_mx("voc.rb" + std::to_string(i * c.n_rb + j) + ".c1.0");
        if (p.dtype == DT_MX) {
            if (prev_planes) mx_in(p, mx_view(vb.pl_nxt, rows_in, ch), ch);            // the previous stage's epilogue wrote lrelu(prev) as planes
            else if (vb.mx_scratch && vb.mx_scratch_bytes >= mx_scratch_bytes(rows_in, ch)) { p.mx_scratch = vb.mx_scratch; p.mx_scratch_size = vb.mx_scratch_bytes; }
            else p.dtype = DT_F32S;
        }
        if (p.dtype == DT_MX && stage_mx) mx_out(p, mx_view(vb.pl_xu, (size_t)rows_out, cout), 0.1f);      // lrelu(x) of models.py:51, shared by the three ResBlocks
        else stage_mx = false;
        // MX stage: the residual stream of a ResBlock exists only as the plane set of lrelu(x, .1) its conv1 reads -- conv2's epilogue rebuilds x from
        // the fp16 hi plane + the fp4 remainder codes (ConvGemmParams::res_x4), so neither the up-conv nor a conv2 inside a ResBlock writes an fp32
        // copy (tools/precision_study_mx.py: 3.4e-4 -> 4.4e-4; 8.7 instead of 14.1 bytes per element and conv2 launch).  Default since round 4 (the
        // round-3 driver run XPASSed every reference fixture through it); ev_config.mx_residual = 1 restores the separate fp32 residual tensor.
        const bool rpl = stage_mx && c.mx_residual == 0;
        if (rpl && !keep) p.out32 = nullptr;                   // (kept stages still get the raw up-conv output: the voc_up tap)
        if (gemm(h, p.dtype == DT_MX ? (p.N == 64 && p.K == 64 ? "voc_conv_c64_mx" : "voc_conv_gemm_mx") : gname, p, valid_in, nullptr, 2.0 / 3.0)) return -1;
        const bool next_up_mx = stage_mx && i + 1 < c.n_up && has_mx("voc.up" + std::to_string(i + 1));
        // MRF sum as partial plane sets (conv_gemm_mx_kernel stages, i.e. >= 128 channels): rb0's last conv writes out_scale * x as fp16 hi plane + fp4 remainder
        // codes, rb1's adds that to its own and rewrites it in place, rb2's adds it and writes the next up-conv's plane set: 2.53 instead of 4 bytes per element
        // and transfer (16 -> 10.1 bytes per stage-output element; tools/precision_study_mx.py: 4.17e-4 -> 4.22e-4 on the zero-mean recipe)
        // (round 6: also at C = 64 -- conv_gemm_mx64_kernel has the same epilogue variants, the fused k = 3 pair writes the partial set itself)
        bool mrf_pl = rpl && !keep && c.mx_mrf == 0 && c.n_rb == 3 && cout % 64 == 0 && next_up_mx && vb.pl_mrf.h;
        for (int j = 0; mrf_pl && j < c.n_rb; ++j) {
            const int k = c.rb_kernels[j];
            mrf_pl = k == 3 || k == 7 || k == 11;
            // C = 64, k = 3: only the fused pair kernel writes / the streamed k = 7 / 11 kernel reads partial sets (conv_c64_mx_kernel does neither), and the pair
            // kernel has no plane-set accumulate-in: the k = 3 ResBlock must be the first of the stage and run fused
            if (mrf_pl && cout == 64 && k == 3)
                mrf_pl = j == 0 && c.fused_pairs == 0 && c.rb_dils[j][c.n_rb_dils - 1] <= 8 && has_wt("voc.rb" + std::to_string(i * c.n_rb + j) + ".c2." + std::to_string(c.n_rb_dils - 1) + ".wcmx");
        }
        U *= s;
        const int shift = ilog2(U);
        const bool last_stage = (i == c.n_up - 1);
        // Optional (ev_config.vocoder_chunk_mb > 0): the ResBlocks of a stage run on row chunks sized for the 256 MB Infinity
        // Cache, so that a chunk's intermediates (xt, the running x of a ResBlock, the fp16 MRF branches) are re-read from the
        // memory-side cache instead of HBM.  The isolated probe (tools/bench_chunked.py) gains 10-15 % on the HBM-bound chains
        // with 2-4 chunks, but in the full forward every chunk size measured LOSES 1-4 % (more launches, each with its own
        // tail: 113-251 instead of 53 conv launches), so the default is whole tensors.
        // Overlapped tiling keeps the result bit-identical: op q of a ResBlock's 6-conv chain runs on the chunk extended by
        // (5 - q) x 256 rows per side, more than any conv's halo (<= 25 rows; the fused pair kernels get the tensor's true
        // bounds so that they zero-pad only at real sequence edges), so the last op's rows [a, b) only ever read
        // rows that were recomputed from valid inputs inside this chunk; what the extensions write outside [a, b) is dead or
        // rewritten by the neighbouring chunk.
        int nchunks = 1;
        if (!keep && !x3 && c.vocoder_chunk_mb > 0) {
            const double tensor_mb = (double)rows_out * cout * 2.0 / 1e6;
            nchunks = (int)(tensor_mb / c.vocoder_chunk_mb + 0.5);
            if (nchunks < 1) nchunks = 1;
            if (nchunks > 64) nchunks = 64;
        }
        const int chunk_rows = ((rows_out + nchunks - 1) / nchunks + 255) / 256 * 256;
        // shift every row-indexed operand of a conv call to the row range [lo, hi)
        auto sub = [&](ConvGemmParams q, int lo, int hi) {
            q.A = (const char*)q.A + (size_t)lo * q.lda * ves;
            if (q.res) q.res = (const char*)q.res + (size_t)lo * q.ldres * (q.res_dtype == DT_F16 ? 2 : 4);
            if (q.acc32) q.acc32 = q.acc32 + (size_t)lo * q.ldacc;
            if (q.add16_a) { q.add16_a = (const char*)q.add16_a + (size_t)lo * q.ldadd * 2; q.add16_b = (const char*)q.add16_b + (size_t)lo * q.ldadd * 2; }
            if (q.out16) q.out16 = (char*)q.out16 + (size_t)lo * q.ldo * 2;
            if (q.out32) q.out32 = q.out32 + (size_t)lo * q.ldo;
            if (q.row_valid) q.row_valid = q.row_valid + (lo >> q.valid_shift);
            q.M = hi - lo;
            return q;
        };
        // The three ResBlocks of a stage only share the stage input and meet again in the MRF sum: the first two run on
        // auxiliary streams beside the third (its last conv waits for both).  A k = 3 chain is HBM-bound and a k = 11 chain
        // MFMA-bound, so their workgroups complement each other on a CU and fill each other's launch tails.  Profiled steps
        // (per-launch events) and chunked execution stay on one stream.  (Split-precision mode: the MRF sum is a running fp32
        // accumulator shared by the three ResBlocks, so they run in order on the handle's stream.)
        const bool conc = c.n_rb == 3 && nchunks == 1 && (!x3 || (mx && !keep && voc_small_batch(Rf))) && !h->profiling && c.vocoder_streams != 1 &&
                          h->aux[0] && h->aux[1];
        // Large batches on the conv_gemm_mx_kernel stages (round 6, ev_config.mx_group == 0): the three ResBlocks advance level by level -- the three conv1 of a
        // pair position, then the three conv2 -- each level ONE grouped launch (launch_conv_gemm_group3: a launch of its own costs every conv 30-50 us of ramp and
        // tail); the last conv2 of each ResBlock stays a launch of its own (the running MRF sum orders them).  Each ResBlock has its own intermediates; descriptors
        // are built in the usual ResBlock order and issued level by level.  Same kernels' code on the same data: the same bits.
        bool grp = stage_mx && rpl && !keep && !conc && nchunks == 1 && c.mx_group == 0 && c.n_rb == 3 && cout % 128 == 0 && vb.pl_t[2].h && vb.pl_a[2].h && vb.pl_b[2].h;
        {
            int seen = 0;
            for (int j = 0; j < c.n_rb; ++j) seen |= c.rb_kernels[j] == 3 ? 1 : (c.rb_kernels[j] == 7 ? 2 : (c.rb_kernels[j] == 11 ? 4 : 8));
            if (seen != 7) grp = false;
        }
        std::vector<ConvGemmParams> pend(grp ? (size_t)c.n_rb * 2 * c.n_rb_dils : 0);
        if (conc) {
            (void)hipEventRecord(h->ev_fork, h->stream);
            (void)hipStreamWaitEvent(h->aux[0], h->ev_fork, 0);
            (void)hipStreamWaitEvent(h->aux[1], h->ev_fork, 0);
        }
        for (int a0 = 0; a0 < rows_out; a0 += chunk_rows) {
        const int b0 = std::min(rows_out, a0 + chunk_rows);
        const double frac = (double)(b0 - a0) / rows_out;
        for (int j = 0; j < c.n_rb; ++j) {
            const int k = c.rb_kernels[j];
            const std::string rb = "voc.rb" + std::to_string(i * c.n_rb + j);
            const void* xcur = vb.xu[i].p;
            const int bj = (conc || grp) ? j : 0;
            hipStream_t sj = (conc && j < 2) ? h->aux[j] : h->stream;
            for (int d = 0; d < c.n_rb_dils; ++d) {
                const int dil = c.rb_dils[j][d];
                const std::string c1 = rb + ".c1." + std::to_string(d), c2 = rb + ".c2." + std::to_string(d);
                // fused pair kernels: C = 32 (every k) and C = 64 with k = 3 (the HBM-bound end of the generator; fp16 mode only)
                const bool fused = !x3 && ((cout == 32 && (k == 3 || k == 7 || k == 11)) || (cout == 64 && k == 3)) && c.fused_pairs == 0;
                // EV_PREC_MX at C = 32: the whole pair in one persistent kernel (ev_pair_mx.h), x fp32 in, fp32 out
                const bool fused_mx = mx && cout == 32 && (k == 3 || k == 7 || k == 11) && (k - 1) * (dil + 1) <= 64 && has_wt(c1 + ".wpmx") && has_wt(c2 + ".wpmx") &&
                                      c.fused_pairs == 0;
                // the plane set of lrelu(x, .1) this pair starts from: conv1's operand and, with rpl, conv2's residual
                const PlaneBuf& xin = d == 0 ? vb.pl_xu : ((d - 1) % 2 == 0 ? vb.pl_a[bj] : vb.pl_b[bj]);
                // EV_PREC_MX at C = 64, k = 3 with the residual in the planes: the pair in one persistent kernel (ev_pair64_mx.h), plane sets in / out -- xt never
                // reaches HBM and the residual comes from the slab conv1 reads (6.1 instead of 14.8 bytes per element; the k = 3 chain of stage 2 is HBM-bound)
                const bool fused_c64 = stage_mx && rpl && cout == 64 && k == 3 && dil <= 8 && c.fused_pairs == 0 && has_wt(c1 + ".wcmx") && has_wt(c2 + ".wcmx");
                if (fused_mx || fused_c64) {
                } else if (stage_mx) {
                    // MX stage: xt only ever exists as conv2's operand planes; x travels as fp32 (the residual) + the planes of lrelu(x); with EV_MX_RESPL=1 as the planes only
                    p = gemm_defaults();
                    if (voc_weights(h, c1, x3, p, true) || p.dtype != DT_MX) return fail(h, "MX stage: %s has no fp4 planes", c1.c_str());
                    mx_in(p, mx_view(xin, (size_t)rows_out, cout), cout);
                    p.M = rows_out; p.N = cout; p.K = cout; p.taps = k; p.dil = dil; p.center = (k - 1) / 2;
                    p.row_valid = h->d_frm_valid; p.valid_shift = shift; p.act = ACT_LRELU; p.act_slope = 0.1f; p.ldo = cout;
                    mx_out(p, mx_view(vb.pl_t[bj], (size_t)rows_out, cout), 1.0f);
                    if (grp) pend[((size_t)j * c.n_rb_dils + d) * 2] = p;
                    else if (gemm(h, cout == 64 ? "voc_conv_c64_mx" : "voc_conv_gemm_mx", p, valid_out, sj)) return -1;
                } else if (!fused) {
                    // xt = lrelu(c1(lrelu(x)))  (models.py:51-53)
                    p = gemm_defaults();
                    if (voc_weights(h, c1, x3, p, false)) return -1;      // (no plane-set input here: DT_MX would only fall back to the split kernel)
                    p.A = xcur; p.lda = cout; p.M = rows_out; p.N = cout; p.K = cout;
                    p.taps = k; p.dil = dil; p.center = (k - 1) / 2; p.row_valid = h->d_frm_valid; p.valid_shift = shift;
                    p.pro_lrelu = 1; p.pro_slope = 0.1f; p.act = ACT_LRELU; p.act_slope = 0.1f; p.ldo = cout;
                    set_out(p, vb.tmp[bj].p);
                    const int e1 = (2 * (c.n_rb_dils - 1 - d) + 1) * 256, lo1 = std::max(0, a0 - e1), hi1 = std::min(rows_out, b0 + e1);
                    if (gemm(h, gname, sub(p, lo1, hi1), valid_out * frac, sj)) return -1;
                }
                // x = c2(xt) + x  (models.py:54-56)
                p = gemm_defaults();
                if (voc_weights(h, c2, x3, p, stage_mx)) return -1;
                p.A = vb.tmp[bj].p; p.lda = cout; p.M = rows_out; p.N = cout; p.K = cout;
                if (stage_mx) {
                    if (p.dtype != DT_MX) return fail(h, "MX stage: %s has no fp4 planes", c2.c_str());
                    mx_in(p, mx_view(vb.pl_t[bj], (size_t)rows_out, cout), cout);
                }
                p.taps = k; p.dil = 1; p.center = (k - 1) / 2; p.row_valid = h->d_frm_valid; p.valid_shift = shift;
                p.res = xcur; p.res_dtype = x3 ? DT_F32 : DT_F16; p.ldres = cout; p.ldo = cout;
                if (rpl) {
                    const MxView xv = mx_view(xin, (size_t)rows_out, cout);
                    p.res = xv.h; p.res_dtype = DT_MX; p.res_x4 = xv.q4[1]; p.res_xs = xv.qs[1]; p.res_xs_stride = xv.qs_stride; p.res_inv_slope = 10.0f;
                }
                if (d + 1 < c.n_rb_dils) {
                    void* dst = (d % 2 == 0) ? vb.rba[bj].p : vb.rbb[bj].p;
                    if (!rpl || keep) set_out(p, dst);          // (rpl: the next pair reads the planes below; the fp32 copy only feeds stage taps)
                    xcur = dst;
                    if (stage_mx) mx_out(p, mx_view(d % 2 == 0 ? vb.pl_a[bj] : vb.pl_b[bj], (size_t)rows_out, cout), 0.1f);     // the next conv1's operand
                } else {
                    // MRF: xs += resblock(x); x = xs / num_kernels (models.py:121-126), then the next leaky_relu
                    // fp16 mode with three ResBlocks (the reference config): the first two scaled branches are kept in fp16 and the
                    // third adds them in its fp32 epilogue (half the HBM traffic of an fp32 running sum; the two extra fp16
                    // roundings are of the size of the one the stage output gets anyway).  Any other count, and the
                    // split-precision mode, use the fp32 running sum.
                    p.out_scale = 1.0f / (float)c.n_rb;
                    const bool mrf16 = (c.n_rb == 3) && !x3;
                    const MxView mv = mrf_pl ? mx_view(vb.pl_mrf, (size_t)rows_out, cout) : MxView{};
                    if (mrf16) {
                        if (j == 2) { p.add16_a = vb.mrf16a.p; p.add16_b = vb.mrf16b.p; p.ldadd = cout; }
                    } else if (mrf_pl) {
                        if (j > 0) { p.acc_h = mv.h; p.acc_x4 = mv.q4[1]; p.acc_xs = mv.qs[1]; p.acc_xs_stride = mv.qs_stride; p.ldacc = cout; }
                    } else if (j > 0) { p.acc32 = (const float*)vb.mrf32.p; p.ldacc = cout; }
                    if (j + 1 < c.n_rb) {
                        if (mrf16) p.out16 = (j == 0) ? vb.mrf16a.p : vb.mrf16b.p;
                        else if (mrf_pl) { mx_out(p, mv, 1.0f); p.mxo_partial = 1; }
                        else p.out32 = (float*)vb.mrf32.p;
                    } else if (x3) {
                        p.out32 = (float*)vb.nxt[i].p;               // raw MRF mean (= the voc_mrf tap); consumers apply the leaky-relu
                        if (next_up_mx) {
                            mx_out(p, mx_view(vb.pl_nxt, (size_t)rows_out, cout), 0.1f);      // ... or read these planes (models.py:118)
                            if (rpl && !keep) p.out32 = nullptr;                              // the next up-conv reads only the planes: no fp32 copy of the stage output
                        }
                    } else {
                        p.post_lrelu = 1; p.post_slope = last_stage ? 0.01f : 0.1f;   // models.py:118 / :127
                        p.out16 = vb.nxt[i].p;
                        if (keep) { p.out32 = (float*)vb.mrf_tap[i].p; p.out32_before_post = 1; }
                    }
                }
                const int e2 = 2 * (c.n_rb_dils - 1 - d) * 256, lo2 = std::max(0, a0 - e2), hi2 = std::min(rows_out, b0 + e2);
                p = sub(p, lo2, hi2);
                if (conc && j == 2 && d + 1 == c.n_rb_dils) {      // the MRF sum reads the other two branches
                    (void)hipStreamWaitEvent(h->stream, h->ev_join[0], 0);
                    (void)hipStreamWaitEvent(h->stream, h->ev_join[1], 0);
                }
                // fp32 running sum (split-precision / mx modes): rb1's last conv adds onto what rb0's wrote (the sum keeps the serial order's bits)
                if (conc && x3 && j == 1 && d + 1 == c.n_rb_dils) (void)hipStreamWaitEvent(h->aux[1], h->ev_join[0], 0);
                if (fused_c64) {
                    WPTR(w1, char, c1 + ".w16"); WPTR(b1, float, c1 + ".b"); WPTR(w1m, char, c1 + ".wcmx");
                    const MxView xv = mx_view(xin, (size_t)rows_out, cout);
                    ResPairParams rp;
                    memset(&rp, 0, sizeof rp);
                    rp.x = xv.h; rp.ldx = cout; rp.w1 = w1; rp.b1 = b1; rp.w2 = p.W; rp.w1_mx = w1m; rp.w2_mx = p.W_mx;
                    rp.M = p.M; rp.k = k; rp.dil = dil; rp.epi = p;
                    rp.epi.mx_x4[0] = xv.q4[0]; rp.epi.mx_x4[1] = xv.q4[1]; rp.epi.mx_xs[0] = xv.qs[0]; rp.epi.mx_xs[1] = xv.qs[1]; rp.epi.mx_xs_stride = xv.qs_stride;
                    const double fl = 2.0 * 2.0 * valid_out * cout * (double)cout * k;
                    ConvGemmParams shape = p; shape.dil = dil;
                    KScope ks(h, "voc_resblock_pair_c64_mx", fl, valid_out * cout * 3.0625 * 2.0, sj, &shape);
                    if (launch_resblock_pair_c64_mx(rp, sj)) return fail(h, "fused MX pair (C = 64): unsupported call (k %d, dil %d)", k, dil);
                } else if (fused_mx) {
                    WPTR(w1, char, c1 + ".w16"); WPTR(b1, float, c1 + ".b"); WPTR(w1m, char, c1 + ".wpmx"); WPTR(w2m, char, c2 + ".wpmx");
                    ResPairParams rp;
                    memset(&rp, 0, sizeof rp);
                    rp.x = p.res; rp.ldx = cout; rp.w1 = w1; rp.b1 = b1; rp.w2 = p.W; rp.w1_mx = w1m; rp.w2_mx = w2m;
                    rp.M = p.M; rp.k = k; rp.dil = dil; rp.gmin = 0; rp.gmax = rows_out; rp.epi = p;
                    if (c.mx_act_format != 0) rp.epi.reserved0 |= 16;          // block-scaled fp4 activation operands (rounds 3-5) instead of E5M2
                    const double fl = 2.0 * 2.0 * valid_out * cout * (double)cout * k;
                    ConvGemmParams shape = p; shape.dil = dil;
                    KScope ks(h, "voc_resblock_pair_c32_mx", fl, valid_out * cout * 4.0 * 2.0, sj, &shape);
                    if (launch_resblock_pair_c32_mx(rp, sj)) return fail(h, "fused MX pair: unsupported call (k %d, dil %d)", k, dil);
                } else if (fused) {
                    // conv1 -> LDS -> conv2 + residual / MRF epilogue in one persistent kernel (ev_gemm.hip)
                    WPTR(w1, char, c1 + ".w16"); WPTR(b1, float, c1 + ".b");
                    ResPairParams rp;
                    memset(&rp, 0, sizeof rp);
                    rp.x = p.res; rp.ldx = cout; rp.w1 = w1; rp.b1 = b1; rp.w2 = p.W; rp.M = p.M; rp.k = k; rp.dil = dil; rp.gmin = -lo2; rp.gmax = rows_out - lo2; rp.epi = p;
                    const double fl = 2.0 * 2.0 * valid_out * frac * cout * (double)cout * k;
                    ConvGemmParams shape = p; shape.dil = dil;
                    KScope ks(h, cout == 32 ? "voc_resblock_pair_c32" : "voc_resblock_pair_c64", fl, valid_out * frac * cout * 2.0 * 2.0, sj, &shape);
                    if (cout == 32) launch_resblock_pair_c32(rp, sj);
                    else launch_resblock_pair_c64(rp, sj);
                } else if (grp) pend[((size_t)j * c.n_rb_dils + d) * 2 + 1] = p;
                else if (gemm(h, p.dtype == DT_MX ? (cout == 64 ? "voc_conv_c64_mx" : "voc_conv_gemm_mx") : gname, p, valid_out * frac, sj)) return -1;
            }
            if (conc && j < 2) (void)hipEventRecord(h->ev_join[j], h->aux[j]);
        }
        if (grp) {
            for (int lvl = 0; lvl < 2 * c.n_rb_dils; ++lvl) {
                ConvGemmParams ps[3];
                for (int j = 0; j < 3; ++j) ps[j] = pend[((size_t)j * c.n_rb_dils) * 2 + lvl];
                if (lvl + 1 < 2 * c.n_rb_dils) {
                    if (gemm_group3(h, "voc_conv_gemm_mx", ps, valid_out)) return -1;
                } else {
                    for (int j = 0; j < 3; ++j)
                        if (gemm(h, "voc_conv_gemm_mx", ps[j], valid_out)) return -1;
                }
            }
        }
        }   // row chunks
        prev = vb.nxt[i].p;
        prev_planes = next_up_mx;
        ch = cout;
    }
    WPTR(wpost, float, "voc.post.w");
    float bpv;
    if (get_scalar(h, "voc.post.b", &bpv)) return -1;
    {
        KScope ks(h, "voc_conv_post", 2.0 * n_frames * U * ch * 7, n_frames * U * (ch * (double)ves + 4.0));
        launch_conv_post(prev, x3 ? 1 : 0, ch, wpost, bpv, 7, x3 ? 0.01f : 1.0f, h->d_frm_valid, ilog2(U), (float*)vb.wavrows.p, Rf * U, ch, h->stream);
    }
    return 0;
}

int total_up(const ev_config& c) { int u = 1; for (int i = 0; i < c.n_up; ++i) u *= c.up_rates[i]; return u; }

void plan_vocoder(ArenaPlan& ap, const ev_config& c, int Rf, bool keep, VocBufs& vb) {
    const bool mx = c.vocoder_precision == EV_PREC_MX;
    const bool x3 = c.vocoder_precision == EV_PREC_X3 || mx;
    const size_t ves = x3 ? 4 : 2;
    vb.pre = ap.rows(Rf, c.up_init_ch, ves);
    if (keep && !x3) vb.pre_tap = ap.rows(Rf, c.up_init_ch, 4);
    int ch = c.up_init_ch, U = 1;
    size_t max_elems = 0;
    for (int i = 0; i < c.n_up; ++i) {
        U *= c.up_rates[i]; ch /= 2;
        max_elems = std::max(max_elems, (size_t)Rf * U * ch);
    }
    // stage buffers are re-used across stages unless taps are kept
    ch = c.up_init_ch; U = 1;
    Buf shared_xu{}, shared_nxt[2];
    for (int i = 0; i < c.n_up; ++i) {
        U *= c.up_rates[i]; ch /= 2;
        if (keep) {
            vb.xu[i] = ap.rows((size_t)Rf * U, ch, ves);
            vb.nxt[i] = ap.rows((size_t)Rf * U, ch, ves);
            if (!x3) vb.mrf_tap[i] = ap.rows((size_t)Rf * U, ch, 4);
        }
    }
    // row pitch differs per stage, so size by elements with the largest pad (C = 256 rows of slack)
    auto mk2 = [&](size_t es) { Buf b; const size_t pad = (size_t)PAD_ROWS * 512 * es; b.bytes = max_elems * es; b.base = ap.take(pad + b.bytes + pad); b.p = ap.dry ? nullptr : b.base + pad; return b; };
    if (!keep) {
        shared_xu = mk2(ves); shared_nxt[0] = mk2(ves); shared_nxt[1] = mk2(ves);
        for (int i = 0; i < c.n_up; ++i) { vb.xu[i] = shared_xu; vb.nxt[i] = shared_nxt[i & 1]; }
    }
    // fp16 mode, and small batches in the mx mode: the three ResBlocks of a stage run concurrently, each with its own intermediates
    const bool per_rb = (c.n_rb == 3) && (!x3 || (mx && !keep && voc_small_batch(Rf)));
    for (int j = 0; j < (per_rb ? 3 : 1); ++j) { vb.tmp[j] = mk2(ves); vb.rba[j] = mk2(ves); vb.rbb[j] = mk2(ves); }
    if (per_rb && !x3) { vb.mrf16a = mk2(2); vb.mrf16b = mk2(2); } else vb.mrf32 = mk2(4);
    vb.wavrows = ap.rows((size_t)Rf * total_up(c), 1, 4);
    if (mx) {
        // plane sets, re-used across the stages that run on the MX kernel (C % 128 == 0): sized by the largest
        size_t hb = 0, qb = 0, sb = 0;
        ch = c.up_init_ch; U = 1;
        for (int i = 0; i < c.n_up; ++i) {
            U *= c.up_rates[i]; ch /= 2;
            if (ch % 64) continue;
            const size_t R = (size_t)Rf * U + 2 * MX_PAD;
            hb = std::max(hb, R * ch * 2); qb = std::max(qb, R * (ch / 2)); sb = std::max(sb, (size_t)std::max(1, ch / 128) * R * 4);
        }
        std::vector<PlaneBuf*> sets = {&vb.pl_xu, &vb.pl_nxt, &vb.pl_mrf};
        // (one set of intermediates per ResBlock also for large batches when their same-level convs are launched grouped: ev_config.mx_group == 0)
        const bool per_rb_planes = per_rb || (c.n_rb == 3 && !keep && c.mx_group == 0 && c.mx_residual == 0);
        for (int j = 0; j < (per_rb_planes ? 3 : 1); ++j) { sets.push_back(&vb.pl_t[j]); sets.push_back(&vb.pl_a[j]); sets.push_back(&vb.pl_b[j]); }
        for (PlaneBuf* b : sets) {
            if (!hb) break;
            b->h = ap.take(hb);
            for (int i = 0; i < 2; ++i) { b->q4[i] = ap.take(qb); b->qs[i] = ap.take(sb); }
        }
        vb.mx_scratch_bytes = (c.up_init_ch % 128 == 0) ? mx_scratch_bytes(Rf, c.up_init_ch) : 0;
        vb.mx_scratch = vb.mx_scratch_bytes ? ap.take(vb.mx_scratch_bytes) : nullptr;
    }
}

// frame layout from mel lengths (host) -> device maps; returns Rf
int build_frame_layout(ev_handle* h, ArenaPlan& ap, bool dry, int B) {
    int64_t rows = GAP;
    h->frm_off.resize(B);
    h->mel_offs.assign(B + 1, 0);
    for (int b = 0; b < B; ++b) {
        h->frm_off[b] = (int32_t)rows;
        rows += h->mel_lens[b] + GAP;
        h->mel_offs[b + 1] = h->mel_offs[b] + h->mel_lens[b];
    }
    h->total_frames = h->mel_offs[B];
    const int Rf = (int)align_up((size_t)rows, ROW_ALIGN);
    h->d_frm_seq = ap.arr<int32_t>(Rf); h->d_frm_pos = ap.arr<int32_t>(Rf); h->d_frm_valid = ap.arr<uint8_t>(Rf);
    h->d_frm_off = ap.arr<int32_t>(B);
    h->d_frm_len = ap.arr<int32_t>(B);
    if (!dry) {
        // only the B first-row offsets and lengths travel; the per-row maps are built on the device (no host loop over rows, no
        // synchronisation: the pinned frame region [PIN_FRAME, ...) is not rewritten before the call's final synchronisation)
        int32_t* off = (int32_t*)(h->pinned + PIN_FRAME); int32_t* len = off + B;
        for (int b = 0; b < B; ++b) { off[b] = h->frm_off[b]; len[b] = h->mel_lens[b]; }
        (void)hipMemcpyAsync(h->d_frm_off, off, (size_t)B * 4, hipMemcpyHostToDevice, h->stream);
        (void)hipMemcpyAsync(h->d_frm_len, len, (size_t)B * 4, hipMemcpyHostToDevice, h->stream);
        launch_row_maps(h->d_frm_off, h->d_frm_len, B, h->d_frm_seq, h->d_frm_pos, h->d_frm_valid, Rf, h->stream);
    }
    return Rf;
}

// gather per-utterance valid rows of a row-layout buffer into a packed fp32 device buffer
int pack_level(ev_handle* h, const void* src, int dtype, int ld, int C, int level_shift, bool token_level, float* dst, int64_t* d_scratch3B) {
    const int B = h->B;
    // host staging lives in the handle (a ring of slots: one call packs at most 4 levels + the stage taps of a test) so that no
    // synchronisation is needed for it to outlive the asynchronous copies
    h->pack_slot = (h->pack_slot + 1) % 8;
    std::vector<int64_t>& host = h->pack_host[h->pack_slot];
    std::vector<int32_t>& rows32 = h->pack_rows[h->pack_slot];
    host.assign(3 * (size_t)B, 0);
    rows32.assign(B, 0);
    int64_t max_rows = 0, out = 0;
    for (int b = 0; b < B; ++b) {
        const int64_t n = token_level ? h->tok_len[b] : ((int64_t)h->mel_lens[b] << level_shift);
        host[b] = token_level ? h->tok_off[b] : ((int64_t)h->frm_off[b] << level_shift);
        host[B + b] = out;
        rows32[b] = (int32_t)n;
        out += n; max_rows = std::max(max_rows, n);
    }
    int64_t* d_row_off = d_scratch3B; int64_t* d_out_off = d_scratch3B + B; int32_t* d_rows = (int32_t*)(d_scratch3B + 2 * B);
    HIPCHK(h, hipMemcpyAsync(d_row_off, host.data(), (size_t)2 * B * 8, hipMemcpyHostToDevice, h->stream));
    HIPCHK(h, hipMemcpyAsync(d_rows, rows32.data(), (size_t)B * 4, hipMemcpyHostToDevice, h->stream));
    launch_pack_rows(src, dtype, ld, C, d_row_off, d_out_off, d_rows, B, max_rows, dst, h->stream);
    return 0;
}

void add_tap(ev_handle* h, const char* name, const void* ptr, int dtype, int ld, int C, int level, int shift) {
    Tap t{ptr, dtype, ld, C, level, shift};
    h->taps[name] = t;
}

}  // namespace

// =================================================================== C ABI

extern "C" {

void ev_default_config(ev_config* c) {
    memset(c, 0, sizeof *c);
    c->abi_version = EV_ABI_VERSION;
    c->n_vocab = 502; c->n_speaker = 2014; c->n_mels = 80; c->hidden = 384; c->heads = 8; c->enc_layers = 4; c->dec_layers = 4;
    c->ffn_kernel = 3; c->bert_dim = 768; c->dur_layers = 2; c->pitch_layers = 3; c->energy_layers = 2; c->var_kernel = 3;
    c->var_embed_kernel = 9; c->n_up = 4;
    const int ur[4] = {8, 8, 2, 2}, uk[4] = {16, 16, 4, 4}, rk[3] = {3, 7, 11}, rd[3] = {1, 3, 5};
    for (int i = 0; i < 4; ++i) { c->up_rates[i] = ur[i]; c->up_kernels[i] = uk[i]; }
    c->up_init_ch = 512; c->n_rb = 3; c->n_rb_dils = 3;
    for (int j = 0; j < 3; ++j) { c->rb_kernels[j] = rk[j]; for (int d = 0; d < 3; ++d) c->rb_dils[j][d] = rd[d]; }
    c->sample_rate = 16000; c->keep_stages = 0; c->token_rate_split = 1;
    // the default IS the contract mode (north_star: waveform within 1e-3 relative L2 of the reference, zero-mean audio included): a caller that follows
    // INTEGRATION.md literally -- ev_default_config, no field overridden -- gets it; fp16 operands (2.4e-3 on zero-mean audio) and strict are opt-ins
    c->decoder_precision = EV_PREC_MX; c->vocoder_precision = EV_PREC_MX;
}

int ev_abi_info(size_t sizes[4]) {
    if (sizes) { sizes[0] = sizeof(ev_config); sizes[1] = sizeof(ev_result); sizes[2] = sizeof(ev_conv_gemm_desc); sizes[3] = sizeof(ev_res_pair_desc); }
    return EV_ABI_VERSION;
}

const char* ev_last_error(ev_handle* h) { return h ? h->err.c_str() : g_create_error.c_str(); }

int ev_create(int device_id, const ev_config* cfg, ev_handle** out) {
    if (!cfg || !out) return fail(nullptr, "ev_create: null argument");
    if (cfg->abi_version != EV_ABI_VERSION) return fail(nullptr, "ev_create: abi_version %d != %d", cfg->abi_version, EV_ABI_VERSION);
    if (cfg->hidden != 384 || cfg->heads != 8) return fail(nullptr, "ev_create: only hidden=384 / heads=8 (d_k=48) kernels are built");
    if (cfg->hidden % 128 || cfg->n_mels > MEL_PAD) return fail(nullptr, "ev_create: unsupported shape");
    if (cfg->n_up < 1 || cfg->n_up > 4 || cfg->n_rb < 1 || cfg->n_rb > 3 || cfg->n_rb_dils < 1 || cfg->n_rb_dils > 4)
        return fail(nullptr, "ev_create: generator layout outside the built range (n_up 1-4, n_rb 1-3, dilations 1-4)");
    for (int i = 0; i < cfg->n_up; ++i)
        if (cfg->up_kernels[i] != 2 * cfg->up_rates[i] || (cfg->up_rates[i] & (cfg->up_rates[i] - 1)))
            return fail(nullptr, "ev_create: upsample stage %d must have kernel = 2*stride and a power-of-two stride", i);
    // shapes the kernels would silently mishandle are rejected here (not discovered as garbage audio):
    //  * conv_post and the last stage exist for 32 channels only; every stage needs C % 32 == 0
    //  * GAP zero rows between utterances must cover every token- / frame-rate conv halo
    //  * a generator conv's span (k-1)*dilation must fit the 64 staged halo rows and the stage's gap rows
    if ((cfg->up_init_ch >> cfg->n_up) != 32 || (cfg->up_init_ch & (cfg->up_init_ch - 1)))
        return fail(nullptr, "ev_create: upsample_initial_channel / 2^n_up must be 32 (got %d / 2^%d)", cfg->up_init_ch, cfg->n_up);
    if ((cfg->ffn_kernel - 1) / 2 > GAP || (cfg->var_embed_kernel - 1) / 2 > GAP || (cfg->var_kernel - 1) / 2 > GAP ||
        !(cfg->ffn_kernel & 1) || !(cfg->var_embed_kernel & 1))
        return fail(nullptr, "ev_create: token / frame-rate conv kernels must be odd and <= %d taps", 2 * GAP + 1);
    {
        int U = 1;
        for (int i = 0; i < cfg->n_up; ++i) {
            U *= cfg->up_rates[i];
            for (int j = 0; j < cfg->n_rb; ++j)
                for (int d = 0; d < cfg->n_rb_dils; ++d) {
                    const int k = cfg->rb_kernels[j], span = (k - 1) * cfg->rb_dils[j][d];
                    if (!(k & 1) || k < 1 || cfg->rb_dils[j][d] < 1 || span > 64 || span / 2 > GAP * U)
                        return fail(nullptr, "ev_create: ResBlock kernel %d / dilation %d at stage %d exceeds the staged halo", k, cfg->rb_dils[j][d], i);
                }
        }
    }
    if (cfg->decoder_precision != EV_PREC_F16 && cfg->decoder_precision != EV_PREC_F32 && cfg->decoder_precision != EV_PREC_X3 &&
        cfg->decoder_precision != EV_PREC_MX)
        return fail(nullptr, "ev_create: unknown decoder_precision %d", cfg->decoder_precision);
    if (cfg->vocoder_precision != EV_PREC_F16 && cfg->vocoder_precision != EV_PREC_X3 && cfg->vocoder_precision != EV_PREC_MX)
        return fail(nullptr, "ev_create: vocoder_precision must be EV_PREC_F16, EV_PREC_X3 or EV_PREC_MX");
    if ((cfg->mx_residual | cfg->decoder_attention | cfg->fused_pairs | cfg->mx_mrf | cfg->decoder_ln_planes | cfg->token_splitk | cfg->mx_act_format | cfg->mx_group) & ~1)
        return fail(nullptr, "ev_create: mx_residual / decoder_attention / fused_pairs / mx_mrf / decoder_ln_planes / token_splitk / mx_act_format / mx_group must be 0 or 1");
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev <= 0) return fail(nullptr, "ev_create: no HIP device available (%s) -- the product path has no CPU fallback", hipGetErrorString(e));
    if (device_id < 0 || device_id >= ndev) return fail(nullptr, "ev_create: device %d out of range (%d devices)", device_id, ndev);
    ev_handle* h = new ev_handle();
    h->cfg = *cfg; h->device = device_id;
    if (hipSetDevice(device_id) != hipSuccess || hipStreamCreateWithFlags(&h->own_stream, hipStreamNonBlocking) != hipSuccess) {
        delete h;
        return fail(nullptr, "ev_create: cannot create stream on device %d", device_id);
    }
    h->stream = h->own_stream;
    if (init_device_kernels(device_id) != 0) {
        (void)hipStreamDestroy(h->own_stream);
        delete h;
        return fail(nullptr, "ev_create: kernel setup failed on device %d (large-LDS opt-in)", device_id);
    }
    for (int j = 0; j < 2; ++j) {
        if (hipStreamCreateWithFlags(&h->aux[j], hipStreamNonBlocking) != hipSuccess) h->aux[j] = nullptr;
        (void)hipEventCreateWithFlags(&h->ev_join[j], hipEventDisableTiming);
    }
    (void)hipEventCreateWithFlags(&h->ev_fork, hipEventDisableTiming);
    *out = h;
    return 0;
}

void ev_destroy(ev_handle* h) {
    if (!h) return;
    (void)hipSetDevice(h->device);
    (void)hipStreamSynchronize(h->stream);
    for (int i = 0; i < 3; ++i) if (h->arena[i]) (void)hipFree(h->arena[i]);
    if (h->sblob) (void)hipFree(h->sblob);
    if (h->pinned) (void)hipHostFree(h->pinned);
    if (h->pe_dev) (void)hipFree(h->pe_dev);
    if (h->wblob && h->wblob_owned) (void)hipFree(h->wblob);
    for (auto e : h->evt_pool) (void)hipEventDestroy(e);
    for (auto& kv : h->region_evt) { (void)hipEventDestroy(kv.second.first); (void)hipEventDestroy(kv.second.second); }
    for (int j = 0; j < 2; ++j) { if (h->aux[j]) (void)hipStreamDestroy(h->aux[j]); if (h->ev_join[j]) (void)hipEventDestroy(h->ev_join[j]); }
    if (h->ev_fork) (void)hipEventDestroy(h->ev_fork);
    if (h->own_stream) (void)hipStreamDestroy(h->own_stream);
    delete h;
}

int ev_set_stream(ev_handle* h, void* s) {
    if (!h) return -1;
    (void)hipStreamSynchronize(h->stream);
    h->stream = s ? (hipStream_t)s : h->own_stream;
    return 0;
}

int ev_load_weights(ev_handle* h, const void* blob, size_t nbytes, const char*) {
    if (!h || !blob) return fail(h, "ev_load_weights: null argument");
    HIPCHK(h, hipSetDevice(h->device));
    if (h->wblob && h->wblob_owned) HIPCHK(h, hipFree(h->wblob));
    h->wblob = nullptr;
    HIPCHK(h, hipMalloc((void**)&h->wblob, nbytes));
    h->wblob_owned = true; h->wbytes = nbytes;
    HIPCHK(h, hipMemcpy(h->wblob, blob, nbytes, hipMemcpyHostToDevice));
    return parse_blob(h, (const char*)blob, nbytes, false);
}

int ev_load_weights_device(ev_handle* h, const void* dptr, size_t nbytes, const char*) {
    if (!h || !dptr) return fail(h, "ev_load_weights_device: null argument");
    HIPCHK(h, hipSetDevice(h->device));
    if (h->wblob && h->wblob_owned) HIPCHK(h, hipFree(h->wblob));
    h->wblob = (char*)dptr; h->wblob_owned = false; h->wbytes = nbytes;
    if (nbytes < 16) return fail(h, "weight blob too small");
    uint32_t count = 0;
    char hdr[16];
    HIPCHK(h, hipMemcpy(hdr, dptr, 16, hipMemcpyDeviceToHost));
    memcpy(&count, hdr + 8, 4);
    const size_t tbl = 16 + (size_t)count * sizeof(BlobEntry);
    if (tbl > nbytes) return fail(h, "weight blob: truncated table");
    std::vector<char> host(tbl);
    HIPCHK(h, hipMemcpy(host.data(), dptr, tbl, hipMemcpyDeviceToHost));
    // parse_blob validates offsets against the full size
    std::vector<char> fake(host);
    return parse_blob(h, fake.data(), nbytes >= tbl ? nbytes : tbl) == 0 ? 0 : -1;
}

int ev_set_forced_durations(ev_handle* h, const int64_t* d, int64_t n) {
    if (!h) return -1;
    h->forced_dur.assign(d, d + n);
    return 0;
}

int ev_memcpy_d2h(ev_handle* h, void* dst, const void* src, size_t n) {
    if (!h || !dst || !src) return fail(h, "ev_memcpy_d2h: null argument");
    HIPCHK(h, hipSetDevice(h->device));
    HIPCHK(h, hipMemcpy(dst, src, n, hipMemcpyDeviceToHost));
    return 0;
}

int ev_set_profiling(ev_handle* h, int enable) { if (!h) return -1; h->profiling = enable != 0; return 0; }
int ev_get_timing(ev_handle* h, const char* name, float* ms) {
    if (!h || !ms) return -1;
    auto it = h->timings.find(name);
    if (it == h->timings.end()) return fail(h, "no timing named %s", name);
    *ms = it->second;
    return 0;
}
int ev_launch_record_count(ev_handle* h) { return h ? (int)h->launches.size() : -1; }
int ev_get_launch_record(ev_handle* h, int idx, ev_launch_record* out) {
    if (!h || !out || idx < 0 || idx >= (int)h->launches.size()) return -1;
    const LaunchRec& r = h->launches[idx];
    memset(out, 0, sizeof *out);
    snprintf(out->name, sizeof out->name, "%s", r.name.c_str());
    out->M = r.M; out->N = r.N; out->K = r.K; out->taps = r.taps; out->dil = r.dil; out->ms = r.ms; out->flops = r.flops; out->bytes = r.bytes;
    return 0;
}
int ev_kernel_stat_count(ev_handle* h) { return h ? (int)h->stats.size() : -1; }
int ev_get_kernel_stat(ev_handle* h, int idx, ev_kernel_stat* out) {
    if (!h || !out || idx < 0 || idx >= (int)h->stats.size()) return -1;
    memset(out, 0, sizeof *out);
    snprintf(out->name, sizeof out->name, "%s", h->stats[idx].name.c_str());
    out->launches = h->stats[idx].launches; out->ms = h->stats[idx].ms; out->flops = h->stats[idx].flops; out->bytes = h->stats[idx].bytes;
    return 0;
}

// ------------------------------------------------------------------- vocoder-only entry
static int finish_wav(ev_handle* h, VocBufs& vb, float* d_wav, int16_t* d_i16, int64_t* d_scr, uint32_t flags, ev_result* out) {
    const int U = total_up(h->cfg);
    if (pack_level(h, vb.wavrows.p, DT_F32, 1, 1, ilog2(U), false, d_wav, d_scr)) return -1;
    if (flags & EV_FLAG_WANT_INT16) launch_wav_to_i16(d_wav, d_i16, h->total_frames * U, h->stream);
    out->wav = d_wav;
    out->wav_i16 = (flags & EV_FLAG_WANT_INT16) ? d_i16 : nullptr;
    out->total_samples = h->total_frames * U;
    return 0;
}

static void register_voc_taps(ev_handle* h, VocBufs& vb) {
    const ev_config& c = h->cfg;
    const bool x3 = c.vocoder_precision != EV_PREC_F16;     // split-precision / MX modes: the stored tensors ARE the raw module outputs
    add_tap(h, "voc_pre", x3 ? vb.pre.p : vb.pre_tap.p, DT_F32, c.up_init_ch, c.up_init_ch, 1, 0);
    int ch = c.up_init_ch, U = 1;
    for (int i = 0; i < c.n_up; ++i) {
        U *= c.up_rates[i]; ch /= 2;
        add_tap(h, ("voc_up" + std::to_string(i)).c_str(), vb.xu[i].p, x3 ? DT_F32 : DT_F16, ch, ch, 2 + i, ilog2(U));
        add_tap(h, ("voc_mrf" + std::to_string(i)).c_str(), x3 ? vb.nxt[i].p : vb.mrf_tap[i].p, DT_F32, ch, ch, 2 + i, ilog2(U));
    }
}

int ev_vocoder(ev_handle* h, int B, const void* mel, int mel_is_f16, const int32_t* mel_lens, uint32_t flags, ev_result* out) {
    if (!h || !mel || !mel_lens || !out || B <= 0) return fail(h, "ev_vocoder: bad argument");
    if (!h->wt.count("voc.post.w")) return fail(h, "ev_vocoder: weights not loaded");
    HIPCHK(h, hipSetDevice(h->device));
    const ev_config& c = h->cfg;
    const bool keep = c.keep_stages != 0;
    const bool voc_x3 = c.vocoder_precision != EV_PREC_F16;      // X3 and MX: fp32 mel rows
    const int U = total_up(c);
    profiling_reset(h);
    h->taps.clear();
    h->B = B; h->total_tokens = 0;
    h->mel_lens.assign(mel_lens, mel_lens + B);
    for (int b = 0; b < B; ++b) if (mel_lens[b] <= 0) return fail(h, "ev_vocoder: mel_lens[%d] = %d", b, mel_lens[b]);
    std::vector<int64_t> elem_off(B);
    int64_t eo = 0;
    for (int b = 0; b < B; ++b) { elem_off[b] = eo; eo += (int64_t)c.n_mels * mel_lens[b]; }
    const size_t es = mel_is_f16 ? 2 : 4;

    Buf mel16; VocBufs vb; float* d_wav = nullptr; int16_t* d_i16 = nullptr; int64_t* d_scr = nullptr; int64_t* d_eoff = nullptr;
    void* d_melin = nullptr; int Rf = 0;
    { int64_t rows = GAP; for (int b = 0; b < B; ++b) rows += mel_lens[b] + GAP; Rf = (int)align_up((size_t)rows, ROW_ALIGN); }
    if ((size_t)B > PIN_MAX_B) return fail(h, "at most %zu utterances per call", PIN_MAX_B);
    if (pinned_reserve(h, PIN_BYTES)) return -1;
    size_t need = 0;
    for (int pass = 0; pass < 2; ++pass) {
        ArenaPlan ap{h, 1, pass == 0};
        if (pass == 1 && arena_reserve(h, 1, need)) return -1;
        build_frame_layout(h, ap, pass == 0, B);
        h->d_mel_len = ap.arr<int32_t>(B);
        d_eoff = ap.arr<int64_t>(B);
        d_scr = ap.arr<int64_t>(3 * (size_t)B + 8);
        if (!(flags & EV_FLAG_DEVICE_INPUTS)) d_melin = ap.take((size_t)eo * es);
        mel16 = ap.rows(Rf, MEL_PAD, voc_x3 ? 4 : 2);      // the generator's input rows (fp32 in the split-precision mode)
        plan_vocoder(ap, c, Rf, keep, vb);
        d_wav = ap.arr<float>((size_t)h->total_frames * U);
        d_i16 = ap.arr<int16_t>((size_t)h->total_frames * U);
        need = ap.off;
    }
    HIPCHK(h, hipMemcpyAsync(h->d_mel_len, mel_lens, (size_t)B * 4, hipMemcpyHostToDevice, h->stream));
    HIPCHK(h, hipMemcpyAsync(d_eoff, elem_off.data(), (size_t)B * 8, hipMemcpyHostToDevice, h->stream));
    const void* melsrc = mel;
    if (!(flags & EV_FLAG_DEVICE_INPUTS)) { HIPCHK(h, hipMemcpyAsync(d_melin, mel, (size_t)eo * es, hipMemcpyHostToDevice, h->stream)); melsrc = d_melin; }
    HIPCHK(h, hipStreamSynchronize(h->stream));
    region_begin(h, "total");
    launch_mel_to_rows(melsrc, mel_is_f16, d_eoff, h->d_frm_seq, h->d_frm_pos, h->d_mel_len, mel16.p, voc_x3 ? 1 : 0, Rf, c.n_mels, MEL_PAD, h->stream);
    region_begin(h, "vocoder");
    if (run_vocoder(h, mel16, Rf, (double)h->total_frames, vb, keep)) return -1;
    HIPCHK(h, hipGetLastError());       // a rejected launch (bad configuration, missing LDS opt-in) must not return stale audio
    region_end(h, "vocoder");
    memset(out, 0, sizeof *out);
    if (finish_wav(h, vb, d_wav, d_i16, d_scr, flags, out)) return -1;
    HIPCHK(h, hipGetLastError());
    region_end(h, "total");
    HIPCHK(h, hipStreamSynchronize(h->stream));
    profiling_collect(h);
    if (keep) register_voc_taps(h, vb);
    out->batch = B; out->total_frames = h->total_frames; out->mel_lens = h->mel_lens.data(); out->mel_offsets = h->mel_offs.data();
    return 0;
}

// ------------------------------------------------------------------- full synthesis
int ev_synthesize(ev_handle* h, int B, const int64_t* ling, const int32_t* cu, const int64_t* speaker, const float* style,
                  const float* content, float alpha, uint32_t flags, ev_result* out) {
    if (!h || !ling || !cu || !speaker || !style || !content || !out || B <= 0) return fail(h, "ev_synthesize: bad argument");
    if (!h->wt.count("tok_emb")) return fail(h, "ev_synthesize: weights not loaded");
    if (cu[0] != 0) return fail(h, "ev_synthesize: cu_seqlens[0] must be 0");
    HIPCHK(h, hipSetDevice(h->device));
    const ev_config& c = h->cfg;
    const int C = c.hidden, U = total_up(c);
    const bool keep = c.keep_stages != 0;
    const bool dev_in = (flags & EV_FLAG_DEVICE_INPUTS) != 0;
    const int dec_prec = c.decoder_precision == EV_PREC_F16 ? DT_F16 : DT_F32;      // X3 and F32 both keep fp32 activations
    const bool voc_x3 = c.vocoder_precision != EV_PREC_F16;      // X3 and MX: fp32 mel rows
    profiling_reset(h);
    h->taps.clear();
    h->B = B;
    const int NT = cu[B];
    h->total_tokens = NT;
    int max_tok = 0;
    h->tok_off.resize(B); h->tok_len.resize(B);
    int64_t rows = GAP;
    for (int b = 0; b < B; ++b) {
        const int n = cu[b + 1] - cu[b];
        if (n <= 0) return fail(h, "ev_synthesize: utterance %d has %d tokens", b, n);
        h->tok_off[b] = (int32_t)rows; h->tok_len[b] = n; rows += n + GAP; max_tok = std::max(max_tok, n);
    }
    const int Rt = (int)align_up((size_t)rows, ROW_ALIGN);
    h->Rt = Rt;
    if ((flags & EV_FLAG_FORCED_DURATIONS) && (int64_t)h->forced_dur.size() != NT) return fail(h, "forced durations: expected %d values", NT);
    if (!dev_in) {      // nn.Embedding raises IndexError on these (model_open_source.py:107,109); device inputs are clamped by the kernels
        for (int j = 0; j < NT; ++j)
            if (ling[j] < 0 || ling[j] >= c.n_vocab) return fail(h, "ev_synthesize: phoneme id %lld at position %d outside [0, %d)", (long long)ling[j], j, c.n_vocab);
        for (int b = 0; b < B; ++b)
            if (speaker[b] < 0 || speaker[b] >= c.n_speaker) return fail(h, "ev_synthesize: speaker id %lld of utterance %d outside [0, %d)", (long long)speaker[b], b, c.n_speaker);
    }
    if (ensure_pe(h, max_tok)) return -1;

    // ---------------- phase 1: token-rate arena
    struct TokBufs {
        Buf x, hb, qkv, ctx, ffn, y, xp, xvar, t1, t2, t1b, t2b, t1c, t2c, pitch, energy, logd, centre;
        std::vector<Buf> ltaps; Buf tokemb_tap;
        int64_t* d_ling; int64_t* d_spk; float* d_style; float* d_content; float* d_u;
        int64_t* d_dur; float* d_logd_packed; float* d_pitch_packed; float* d_energy_packed; int64_t* d_forced; int64_t* d_scr;
    } tb;
    size_t tok_arena_end = 0;
    for (int pass = 0; pass < 2; ++pass) {
        ArenaPlan ap{h, 0, pass == 0};
        if (pass == 1 && arena_reserve(h, 0, tok_arena_end)) return -1;
        h->d_tok_seq = ap.arr<int32_t>(Rt); h->d_tok_pos = ap.arr<int32_t>(Rt); h->d_tok_valid = ap.arr<uint8_t>(Rt);
        h->d_tok_off = ap.arr<int32_t>(B); h->d_tok_len = ap.arr<int32_t>(B); h->d_cu = ap.arr<int32_t>(B + 1);
        h->d_mel_len = ap.arr<int32_t>(B);
        tb.d_ling = ap.arr<int64_t>(NT); tb.d_spk = ap.arr<int64_t>(B); tb.d_style = ap.arr<float>((size_t)B * c.bert_dim);
        tb.d_content = ap.arr<float>((size_t)B * c.bert_dim); tb.d_u = ap.arr<float>((size_t)B * C);
        tb.d_dur = ap.arr<int64_t>(NT); tb.d_logd_packed = ap.arr<float>(NT); tb.d_pitch_packed = ap.arr<float>(NT);
        tb.d_energy_packed = ap.arr<float>(NT); tb.d_forced = ap.arr<int64_t>(NT); tb.d_scr = ap.arr<int64_t>(3 * (size_t)B + 8);
        tb.x = ap.rows(Rt, C, 4); tb.hb = ap.rows(Rt, C, 4); tb.qkv = ap.rows(Rt, 3 * C, 4); tb.ctx = ap.rows(Rt, C, 4);
        tb.ffn = ap.rows(Rt, 4 * C, 4); tb.y = ap.rows(Rt, C, 4); tb.xp = ap.rows(Rt, C, 4); tb.xvar = ap.rows(Rt, C, 4);
        tb.t1 = ap.rows(Rt, C, 4); tb.t2 = ap.rows(Rt, C, 4);
        tb.t1b = ap.rows(Rt, C, 4); tb.t2b = ap.rows(Rt, C, 4); tb.t1c = ap.rows(Rt, C, 4); tb.t2c = ap.rows(Rt, C, 4);
        tb.pitch = ap.rows(Rt, 1, 4); tb.energy = ap.rows(Rt, 1, 4); tb.logd = ap.rows(Rt, 1, 4); tb.centre = ap.rows(Rt, 1, 4);
        if (keep) { tb.ltaps.resize(c.enc_layers); for (auto& b : tb.ltaps) b = ap.rows(Rt, C, 4); tb.tokemb_tap = ap.rows(Rt, C, 4); }
        {          // split-K partial sums (tok_splitk): 4 ranges x hidden columns
            const Buf kb = (c.token_splitk == 0 && c.token_rate_split != 0) ? ap.rows(Rt, 4 * C, 4) : Buf{};
            h->tok_ks = kb.p; h->tok_ks_bytes = kb.p ? (size_t)Rt * 4 * C * 4 : 0;
        }
        tok_arena_end = ap.off;
    }
    // token layout: B offsets / lengths / cu_seqlens through the pinned token region, per-row maps built on the device
    if ((size_t)B > PIN_MAX_B) return fail(h, "ev_synthesize: at most %zu utterances per call", PIN_MAX_B);
    if (pinned_reserve(h, PIN_BYTES)) return -1;
    {
        int32_t* off = (int32_t*)h->pinned; int32_t* len = off + B; int32_t* pcu = len + B;
        for (int b = 0; b < B; ++b) { off[b] = h->tok_off[b]; len[b] = h->tok_len[b]; }
        memcpy(pcu, cu, (size_t)(B + 1) * 4);
        HIPCHK(h, hipMemcpyAsync(h->d_tok_off, off, (size_t)B * 4, hipMemcpyHostToDevice, h->stream));
        HIPCHK(h, hipMemcpyAsync(h->d_tok_len, len, (size_t)B * 4, hipMemcpyHostToDevice, h->stream));
        HIPCHK(h, hipMemcpyAsync(h->d_cu, pcu, (size_t)(B + 1) * 4, hipMemcpyHostToDevice, h->stream));
        launch_row_maps(h->d_tok_off, h->d_tok_len, B, h->d_tok_seq, h->d_tok_pos, h->d_tok_valid, Rt, h->stream);
        // caller-owned inputs: borrowed for the duration of the call (host pointers are pageable: the runtime stages them before
        // hipMemcpyAsync returns; device pointers are read in stream order)
        const hipMemcpyKind kind = dev_in ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice;
        HIPCHK(h, hipMemcpyAsync(tb.d_ling, ling, (size_t)NT * 8, kind, h->stream));
        HIPCHK(h, hipMemcpyAsync(tb.d_spk, speaker, (size_t)B * 8, kind, h->stream));
        HIPCHK(h, hipMemcpyAsync(tb.d_style, style, (size_t)B * c.bert_dim * 4, kind, h->stream));
        HIPCHK(h, hipMemcpyAsync(tb.d_content, content, (size_t)B * c.bert_dim * 4, kind, h->stream));
        if (flags & EV_FLAG_FORCED_DURATIONS)
            HIPCHK(h, hipMemcpyAsync(tb.d_forced, h->forced_dur.data(), (size_t)NT * 8, hipMemcpyHostToDevice, h->stream));
    }
    region_begin(h, "total");
    region_begin(h, "am");
    region_begin(h, "encoder");
    RowCtx trc{Rt, h->d_tok_valid, h->d_tok_seq, h->d_tok_off, h->d_tok_len, B, max_tok, (double)NT};
    WPTR(tok_emb, float, "tok_emb"); WPTR(spk_emb, float, "spk_emb");
    float alphas[2];
    if (get_scalar(h, "enc.alpha", &alphas[0]) || get_scalar(h, "dec.alpha", &alphas[1])) return -1;
    { KScope ks(h, "embed_pe", 0, (double)NT * C * 12.0);
      launch_embed_pe(tb.d_ling, h->d_cu, h->d_tok_seq, h->d_tok_pos, tok_emb, c.n_vocab, h->pe_dev, alphas[0], (float*)tb.x.p, keep ? (float*)tb.tokemb_tap.p : nullptr, Rt, C, h->stream); }
    if (run_stack(h, "enc", c.enc_layers, DT_F32, trc, tb.x, tb.hb, tb.qkv, tb.ctx, tb.ffn, tb.y, nullptr, keep ? &tb.ltaps : nullptr)) return -1;
    HIPCHK(h, hipGetLastError());
    region_end(h, "encoder");
    region_begin(h, "variance");
    // embed_projection1 (model_open_source.py:109-111): time-varying part as a GEMM, conditioning part as a per-utterance vector
    WPTR(wcond, float, "proj.wcond"); WPTR(bproj, float, "proj.b"); WPTR(wproj, char, "proj.w32");
    { KScope ks(h, "cond_vector", 2.0 * B * C * (C + 2.0 * c.bert_dim), 0);
      launch_cond_vector(tb.d_spk, tb.d_style, tb.d_content, spk_emb, c.n_speaker, wcond, bproj, tb.d_u, B, C, c.bert_dim, h->stream); }
    {
        ConvGemmParams p = gemm_defaults();
        p.dtype = DT_F32; p.A = tb.y.p; p.lda = C; p.W = wproj; p.M = Rt; p.N = C; p.K = C; p.row_valid = h->d_tok_valid;
        p.row_seq = h->d_tok_seq; p.seq_bias = tb.d_u; p.ld_seq_bias = C; p.out32 = (float*)tb.xp.p; p.ldo = C;
        if (tok_weights(h, "proj.w", p)) return -1;
        if (gemm(h, "variance_f32_gemm", p, NT)) return -1;
    }
    {
        // the three predictors only share their input: pitch and energy run on the auxiliary streams beside the duration
        // predictor (each conv is one partial wave of 396 workgroups on 256 CUs)
        const bool conc = !h->profiling && c.vocoder_streams != 1 && h->aux[0] && h->aux[1];
        if (conc) {
            (void)hipEventRecord(h->ev_fork, h->stream);
            (void)hipStreamWaitEvent(h->aux[0], h->ev_fork, 0);
            (void)hipStreamWaitEvent(h->aux[1], h->ev_fork, 0);
        }
        if (run_predictor(h, "pitch", c.pitch_layers, trc, tb.xp, tb.t1b, tb.t2b, (float*)tb.pitch.p, conc ? h->aux[0] : nullptr)) return -1;
        if (run_predictor(h, "energy", c.energy_layers, trc, tb.xp, tb.t1c, tb.t2c, (float*)tb.energy.p, conc ? h->aux[1] : nullptr)) return -1;
        if (conc) { (void)hipEventRecord(h->ev_join[0], h->aux[0]); (void)hipEventRecord(h->ev_join[1], h->aux[1]); }
        if (run_predictor(h, "dur", c.dur_layers, trc, tb.xp, tb.t1, tb.t2, (float*)tb.logd.p)) return -1;
        if (conc) { (void)hipStreamWaitEvent(h->stream, h->ev_join[0], 0); (void)hipStreamWaitEvent(h->stream, h->ev_join[1], 0); }
    }
    {
        WPTR(wp, float, "pitch_emb.w"); WPTR(bp, float, "pitch_emb.b"); WPTR(we, float, "energy_emb.w"); WPTR(be, float, "energy_emb.b");
        KScope ks(h, "var_embed_add", 0, (double)NT * C * 8.0);
        launch_var_embed_add((const float*)tb.xp.p, (const float*)tb.pitch.p, (const float*)tb.energy.p, wp, bp, we, be, h->d_tok_valid,
                             (float*)tb.xvar.p, Rt, C, c.var_embed_kernel, h->stream);
    }
    { KScope ks(h, "durations", 0, 0);
      launch_durations((const float*)tb.logd.p, h->d_tok_off, h->d_tok_len, B, alpha, (flags & EV_FLAG_FORCED_DURATIONS) ? tb.d_forced : nullptr,
                       h->d_cu, tb.d_dur, tb.d_logd_packed, (float*)tb.centre.p, h->d_mel_len, h->stream); }
    HIPCHK(h, hipGetLastError());
    region_end(h, "variance");
    // the reference has the same host sync here (alignment.py:195 `.item()`)
    h->mel_lens.resize(B);
    HIPCHK(h, hipMemcpyAsync(h->mel_lens.data(), h->d_mel_len, (size_t)B * 4, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    int max_frames = 0;
    for (int b = 0; b < B; ++b) {
        if (h->mel_lens[b] <= 0) return fail(h, "utterance %d produced %d mel frames", b, h->mel_lens[b]);
        max_frames = std::max(max_frames, h->mel_lens[b]);
    }
    if (ensure_pe(h, max_frames)) return -1;
    const float* pe = h->pe_dev;

    // ---------------- phase 2: frame-rate arena (placed after the token arena)
    struct FrmBufs { Buf x, hb, qkv, ctx, ffn, y, mel32, mel16, up_tap, y_tap; std::vector<Buf> ltaps; float* d_mel; float* d_wav; int16_t* d_i16; } fb;
    VocBufs vb;
    int Rf = 0;
    const size_t esd = dec_prec == DT_F16 ? 2 : 4;
    const bool dec_mx = c.decoder_precision == EV_PREC_MX && C % 128 == 0;
    DecMx dmx{};
    { int64_t r = GAP; for (int b = 0; b < B; ++b) r += h->mel_lens[b] + GAP; Rf = (int)align_up((size_t)r, ROW_ALIGN); }
    if ((size_t)B > PIN_MAX_B) return fail(h, "at most %zu utterances per call", PIN_MAX_B);
    if (pinned_reserve(h, PIN_BYTES)) return -1;
    size_t frm_need = 0;
    for (int pass = 0; pass < 2; ++pass) {
        ArenaPlan ap{h, 1, pass == 0};
        if (pass == 1 && arena_reserve(h, 1, frm_need)) return -1;
        build_frame_layout(h, ap, pass == 0, B);
        fb.x = ap.rows(Rf, C, 4); fb.hb = ap.rows(Rf, C, esd); fb.qkv = ap.rows(Rf, 3 * C, esd); fb.ctx = ap.rows(Rf, C, esd);
        fb.ffn = ap.rows(Rf, 4 * C, esd); fb.y = ap.rows(Rf, C, esd); fb.mel32 = ap.rows(Rf, MEL_PAD, 4); fb.mel16 = ap.rows(Rf, MEL_PAD, 2);
        if (dec_mx) {
            const size_t R = (size_t)Rf + 2 * MX_PAD, F = 4 * (size_t)C;
            dmx.ffn.h = ap.take(R * F * 2);
            for (int i = 0; i < 2; ++i) { dmx.ffn.q4[i] = ap.take(R * F / 2); dmx.ffn.qs[i] = ap.take(F / 128 * R * 4); }
            dmx.scratch_bytes = mx_scratch_bytes(Rf, C);
            dmx.scratch = ap.take(dmx.scratch_bytes);
        }
        if (keep) { fb.ltaps.resize(c.dec_layers); for (auto& b : fb.ltaps) b = ap.rows(Rf, C, 4); fb.up_tap = ap.rows(Rf, C, 4); fb.y_tap = ap.rows(Rf, C, 4); }
        fb.d_mel = ap.arr<float>((size_t)h->total_frames * c.n_mels);
        if (!(flags & EV_FLAG_NO_VOCODER)) {
            plan_vocoder(ap, c, Rf, keep, vb);
            fb.d_wav = ap.arr<float>((size_t)h->total_frames * U);
            fb.d_i16 = ap.arr<int16_t>((size_t)h->total_frames * U);
        }
        frm_need = ap.off;
    }
    h->Rf = Rf;
    region_begin(h, "decoder");
    RowCtx frc{Rf, h->d_frm_valid, h->d_frm_seq, h->d_frm_off, h->d_mel_len, B, max_frames, (double)h->total_frames};
    { KScope ks(h, "gauss_upsample", 0, (double)h->total_frames * C * 8.0);
      launch_gauss_upsample((const float*)tb.xvar.p, (const float*)tb.centre.p, h->d_tok_off, h->d_tok_len, h->d_frm_seq, h->d_frm_pos, pe,
                            alphas[1], 0.1f, (float*)fb.x.p, keep ? (float*)fb.up_tap.p : nullptr, Rf, C, h->stream); }
    if (run_stack(h, "dec", c.dec_layers, dec_prec, frc, fb.x, fb.hb, fb.qkv, fb.ctx, fb.ffn, fb.y,
                  (keep && dec_prec == DT_F16) ? (float*)fb.y_tap.p : nullptr, keep ? &fb.ltaps : nullptr, dec_mx ? &dmx : nullptr)) return -1;
    {
        WPTR(wm, char, dec_prec == DT_F16 ? "to_mel.w16" : "to_mel.w32"); WPTR(bm, float, "to_mel.b");
        ConvGemmParams p = gemm_defaults();
        p.dtype = dec_prec; p.A = fb.y.p; p.lda = C; p.W = wm; p.bias = bm; p.M = Rf; p.N = MEL_PAD; p.K = C; p.row_valid = h->d_frm_valid;
        p.out32 = (float*)fb.mel32.p; p.ldo = MEL_PAD;
        if (!voc_x3) p.out16 = fb.mel16.p;            // the fp16 generator reads fp16 mel rows, the split-precision one the fp32 rows
        if (dec_prec == DT_F32 && tok_weights(h, "to_mel.w", p)) return -1;
        if (gemm(h, dec_prec == DT_F16 ? "dec_f16_gemm" : "dec_f32_gemm", p, (double)h->total_frames)) return -1;
    }
    HIPCHK(h, hipGetLastError());
    region_end(h, "decoder");
    region_end(h, "am");
    // packed outputs
    if (pack_level(h, fb.mel32.p, DT_F32, MEL_PAD, c.n_mels, 0, false, fb.d_mel, tb.d_scr)) return -1;
    if (pack_level(h, tb.pitch.p, DT_F32, 1, 1, 0, true, tb.d_pitch_packed, tb.d_scr)) return -1;
    if (pack_level(h, tb.energy.p, DT_F32, 1, 1, 0, true, tb.d_energy_packed, tb.d_scr)) return -1;
    memset(out, 0, sizeof *out);
    if (!(flags & EV_FLAG_NO_VOCODER)) {
        region_begin(h, "vocoder");
        if (run_vocoder(h, voc_x3 ? fb.mel32 : fb.mel16, Rf, (double)h->total_frames, vb, keep)) return -1;
        HIPCHK(h, hipGetLastError());
        region_end(h, "vocoder");
        if (finish_wav(h, vb, fb.d_wav, fb.d_i16, tb.d_scr, flags, out)) return -1;
        HIPCHK(h, hipGetLastError());
    }
    region_end(h, "total");
    HIPCHK(h, hipStreamSynchronize(h->stream));
    profiling_collect(h);
    if (keep) {
        add_tap(h, "tok_emb", tb.tokemb_tap.p, DT_F32, C, C, 0, 0);
        for (int i = 0; i < c.enc_layers; ++i) add_tap(h, ("enc_l" + std::to_string(i)).c_str(), tb.ltaps[i].p, DT_F32, C, C, 0, 0);
        add_tap(h, "enc_out", tb.y.p, DT_F32, C, C, 0, 0);
        add_tap(h, "x_proj", tb.xp.p, DT_F32, C, C, 0, 0);
        add_tap(h, "x_var", tb.xvar.p, DT_F32, C, C, 0, 0);
        add_tap(h, "upsampled", fb.up_tap.p, DT_F32, C, C, 1, 0);
        for (int i = 0; i < c.dec_layers; ++i) add_tap(h, ("dec_l" + std::to_string(i)).c_str(), fb.ltaps[i].p, DT_F32, C, C, 1, 0);
        if (dec_prec == DT_F16) add_tap(h, "dec_out", fb.y_tap.p, DT_F32, C, C, 1, 0);
        else add_tap(h, "dec_out", fb.y.p, DT_F32, C, C, 1, 0);
        add_tap(h, "mel", fb.mel32.p, DT_F32, MEL_PAD, c.n_mels, 1, 0);
        if (!(flags & EV_FLAG_NO_VOCODER)) register_voc_taps(h, vb);
    }
    out->batch = B; out->total_tokens = NT; out->total_frames = h->total_frames;
    out->mel = fb.d_mel; out->durations = tb.d_dur; out->log_durations = tb.d_logd_packed; out->pitch = tb.d_pitch_packed;
    out->energy = tb.d_energy_packed; out->mel_lens = h->mel_lens.data(); out->mel_offsets = h->mel_offs.data();
    h->last_dur = tb.d_dur;
    return 0;
}

// ------------------------------------------------------------------- SimBERT prompt / content encoder
void ev_default_bert_config(ev_bert_config* c) {
    memset(c, 0, sizeof *c);
    c->vocab_size = 13685; c->hidden = 768; c->layers = 12; c->heads = 12; c->intermediate = 3072; c->max_position = 512;
    c->type_vocab = 2; c->ln_eps = 1e-12f;
}

int ev_style_load_weights(ev_handle* h, const ev_bert_config* cfg, const void* blob, size_t nbytes) {
    if (!h || !cfg || !blob) return fail(h, "ev_style_load_weights: null argument");
    if (cfg->hidden % 128 || cfg->hidden > 1024 || cfg->heads <= 0 || cfg->hidden / cfg->heads != 64 || cfg->intermediate % 64 || cfg->layers <= 0)
        return fail(h, "ev_style_load_weights: only hidden %% 128 == 0 (<= 1024) with 64-wide heads is built (BERT-base: 768 / 12)");
    HIPCHK(h, hipSetDevice(h->device));
    if (h->sblob) { HIPCHK(h, hipStreamSynchronize(h->stream)); HIPCHK(h, hipFree(h->sblob)); h->sblob = nullptr; }
    h->style_loaded = false;
    HIPCHK(h, hipMalloc((void**)&h->sblob, nbytes));
    h->sbytes = nbytes;
    HIPCHK(h, hipMemcpy(h->sblob, blob, nbytes, hipMemcpyHostToDevice));
    if (parse_blob(h, (const char*)blob, nbytes, true)) return -1;
    const WeightEntry* we = W(h, "sb.emb.word");
    if (!we) return -1;
    if ((int)we->dims[0] != cfg->vocab_size || (int)we->dims[1] != cfg->hidden) return fail(h, "ev_style_load_weights: word embedding %llu x %llu does not match the config", (unsigned long long)we->dims[0], (unsigned long long)we->dims[1]);
    h->bcfg = *cfg;
    h->style_loaded = true;
    return 0;
}

// BertModel.forward -> pooler_output for B texts packed back to back (reference simbert.py:49-55 through
// inference_am_vocoder_joint.py:25-38, which tokenises one text per call: attention_mask all ones; here each text attends to its
// own tokens only, the same B = 1 semantics).  fp32-class arithmetic throughout (split-precision GEMMs, exact-fp32 MFMA
// attention): the pooled output conditions the duration predictor, whose integer output must stay bit-exact.
int ev_style_embed(ev_handle* h, int B, const int64_t* input_ids, const int64_t* token_type_ids, const int32_t* cu, uint32_t flags, float* out) {
    if (!h || !input_ids || !cu || !out || B <= 0) return fail(h, "ev_style_embed: bad argument");
    if (!h->style_loaded) return fail(h, "ev_style_embed: ev_style_load_weights first");
    if (cu[0] != 0) return fail(h, "ev_style_embed: cu_seqlens[0] must be 0");
    HIPCHK(h, hipSetDevice(h->device));
    const ev_bert_config& bc = h->bcfg;
    const int H = bc.hidden, I = bc.intermediate;
    const bool dev_in = (flags & EV_FLAG_DEVICE_INPUTS) != 0;
    const int NT = cu[B];
    std::vector<int32_t> off(B), len(B);
    int64_t rows = GAP; int max_len = 0;
    for (int b = 0; b < B; ++b) {
        const int n = cu[b + 1] - cu[b];
        if (n <= 0) return fail(h, "ev_style_embed: text %d has %d tokens", b, n);
        if (n > bc.max_position) return fail(h, "ev_style_embed: text %d has %d tokens > max_position_embeddings %d", b, n, bc.max_position);
        off[b] = (int32_t)rows; len[b] = n; rows += n + GAP; max_len = std::max(max_len, n);
    }
    if (!dev_in)
        for (int j = 0; j < NT; ++j)
            if (input_ids[j] < 0 || input_ids[j] >= bc.vocab_size) return fail(h, "ev_style_embed: token id %lld at position %d outside [0, %d)", (long long)input_ids[j], j, bc.vocab_size);
    const int Rt = (int)align_up((size_t)rows, ROW_ALIGN);
    Buf x, t, qkv, ctx, ffn; int32_t *d_seq, *d_pos, *d_off, *d_len, *d_cu; uint8_t* d_valid; int64_t *d_ids, *d_tt; float* d_out;
    size_t need = 0;
    for (int pass = 0; pass < 2; ++pass) {
        ArenaPlan ap{h, 2, pass == 0};
        if (pass == 1 && arena_reserve(h, 2, need)) return -1;
        d_seq = ap.arr<int32_t>(Rt); d_pos = ap.arr<int32_t>(Rt); d_valid = ap.arr<uint8_t>(Rt);
        d_off = ap.arr<int32_t>(B); d_len = ap.arr<int32_t>(B); d_cu = ap.arr<int32_t>(B + 1);
        d_ids = ap.arr<int64_t>(NT); d_tt = ap.arr<int64_t>(NT); d_out = ap.arr<float>((size_t)B * H);
        x = ap.rows(Rt, H, 4); t = ap.rows(Rt, H, 4); qkv = ap.rows(Rt, 3 * H, 4); ctx = ap.rows(Rt, H, 4); ffn = ap.rows(Rt, I, 4);
        need = ap.off;
    }
    if ((size_t)B > PIN_MAX_B) return fail(h, "ev_style_embed: at most %zu texts per call", PIN_MAX_B);
    if (pinned_reserve(h, PIN_BYTES)) return -1;
    {
        int32_t* poff = (int32_t*)h->pinned; int32_t* plen = poff + B; int32_t* pcu = plen + B;
        for (int b = 0; b < B; ++b) { poff[b] = off[b]; plen[b] = len[b]; }
        memcpy(pcu, cu, (size_t)(B + 1) * 4);
        HIPCHK(h, hipMemcpyAsync(d_off, poff, (size_t)B * 4, hipMemcpyHostToDevice, h->stream));
        HIPCHK(h, hipMemcpyAsync(d_len, plen, (size_t)B * 4, hipMemcpyHostToDevice, h->stream));
        HIPCHK(h, hipMemcpyAsync(d_cu, pcu, (size_t)(B + 1) * 4, hipMemcpyHostToDevice, h->stream));
        launch_row_maps(d_off, d_len, B, d_seq, d_pos, d_valid, Rt, h->stream);
        const hipMemcpyKind kind = dev_in ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice;
        HIPCHK(h, hipMemcpyAsync(d_ids, input_ids, (size_t)NT * 8, kind, h->stream));
        if (token_type_ids) HIPCHK(h, hipMemcpyAsync(d_tt, token_type_ids, (size_t)NT * 8, kind, h->stream));
    }
    WPTR(wword, float, "sb.emb.word"); WPTR(wpos, float, "sb.emb.pos"); WPTR(wtype, float, "sb.emb.type");
    WPTR(eg, float, "sb.emb.ln.g"); WPTR(eb, float, "sb.emb.ln.b");
    launch_bert_embed(d_ids, token_type_ids ? d_tt : nullptr, d_cu, d_seq, d_pos, wword, wpos, wtype, bc.vocab_size, bc.max_position, bc.type_vocab,
                      (float*)t.p, Rt, H, h->stream);
    LayerNormParams ln{};
    ln.ldx = H; ln.rows = Rt; ln.C = H; ln.eps = bc.ln_eps; ln.row_valid = d_valid; ln.ldo = H;
    ln.x = (const float*)t.p; ln.gamma = eg; ln.beta = eb; ln.out32 = (float*)x.p;
    launch_layernorm(ln, h->stream);
    auto bert_gemm = [&](const std::string& base, ConvGemmParams& p) -> int {
        const WeightEntry* hi = W(h, base + ".w16"); const WeightEntry* lo = W(h, base + ".w32l"); const WeightEntry* bias = W(h, base + ".b");
        if (!hi || !lo || !bias) return -1;
        p.dtype = DT_F32S; p.W = hi->ptr; p.W_lo = lo->ptr; p.bias = reinterpret_cast<const float*>(bias->ptr);
        p.M = Rt; p.row_valid = d_valid;
        return gemm(h, "style_gemm", p, (double)NT);
    };
    for (int i = 0; i < bc.layers; ++i) {
        const std::string lp = "sb." + std::to_string(i);
        WPTR(g1, float, lp + ".ln1.g"); WPTR(b1, float, lp + ".ln1.b"); WPTR(g2, float, lp + ".ln2.g"); WPTR(b2, float, lp + ".ln2.b");
        ConvGemmParams p = gemm_defaults();
        p.A = x.p; p.lda = H; p.N = 3 * H; p.K = H; p.out32 = (float*)qkv.p; p.ldo = 3 * H;
        if (bert_gemm(lp + ".qkv", p)) return -1;
        AttnParams ap{};
        ap.qkv = qkv.p; ap.dtype = DT_F32; ap.ld = 3 * H; ap.C = H; ap.heads = bc.heads; ap.seq_off = d_off; ap.seq_len = d_len; ap.B = B;
        ap.max_len = max_len; ap.out = ctx.p; ap.ldo = H;
        launch_attention(ap, h->stream);
        p = gemm_defaults();           // BertSelfOutput: LayerNorm(dense(ctx) + x)
        p.A = ctx.p; p.lda = H; p.N = H; p.K = H; p.res = x.p; p.res_dtype = DT_F32; p.ldres = H; p.out32 = (float*)t.p; p.ldo = H;
        if (bert_gemm(lp + ".out", p)) return -1;
        ln.x = (const float*)t.p; ln.gamma = g1; ln.beta = b1; ln.out32 = (float*)x.p;
        launch_layernorm(ln, h->stream);
        p = gemm_defaults();           // BertIntermediate: gelu(dense(x)) (erf form)
        p.A = x.p; p.lda = H; p.N = I; p.K = H; p.act = ACT_GELU; p.out32 = (float*)ffn.p; p.ldo = I;
        if (bert_gemm(lp + ".ffn1", p)) return -1;
        p = gemm_defaults();           // BertOutput: LayerNorm(dense(h) + x)
        p.A = ffn.p; p.lda = I; p.N = H; p.K = I; p.res = x.p; p.res_dtype = DT_F32; p.ldres = H; p.out32 = (float*)t.p; p.ldo = H;
        if (bert_gemm(lp + ".ffn2", p)) return -1;
        ln.x = (const float*)t.p; ln.gamma = g2; ln.beta = b2; ln.out32 = (float*)x.p;
        launch_layernorm(ln, h->stream);
    }
    WPTR(pw, float, "sb.pool.w"); WPTR(pb, float, "sb.pool.b");
    launch_bert_pooler((const float*)x.p, H, d_off, pw, pb, d_out, B, H, h->stream);
    HIPCHK(h, hipGetLastError());
    HIPCHK(h, hipMemcpyAsync(out, d_out, (size_t)B * H * 4, dev_in ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return 0;
}

int64_t ev_get_stage(ev_handle* h, const char* name, void* host_dst, size_t cap) {
    if (!h || !name) return -1;
    HIPCHK(h, hipSetDevice(h->device));
    if (!strcmp(name, "dur")) {
        const size_t need = (size_t)h->total_tokens * 8;
        if (!host_dst) return (int64_t)need;
        if (cap < need || !h->last_dur) return fail(h, "ev_get_stage(dur): buffer too small or no synthesis yet");
        HIPCHK(h, hipMemcpy(host_dst, h->last_dur, need, hipMemcpyDeviceToHost));
        return (int64_t)need;
    }
    if (!strcmp(name, "mel_len")) {
        const size_t need = (size_t)h->B * 8;
        if (!host_dst) return (int64_t)need;
        if (cap < need) return fail(h, "ev_get_stage(mel_len): buffer too small");
        for (int b = 0; b < h->B; ++b) ((int64_t*)host_dst)[b] = h->mel_lens[b];
        return (int64_t)need;
    }
    auto it = h->taps.find(name);
    if (it == h->taps.end()) return fail(h, "ev_get_stage: unknown stage '%s' (keep_stages=%d)", name, h->cfg.keep_stages);
    const Tap& t = it->second;
    int64_t nrows = 0;
    for (int b = 0; b < h->B; ++b) nrows += t.level == 0 ? h->tok_len[b] : ((int64_t)h->mel_lens[b] << t.shift);
    const size_t need = (size_t)nrows * t.C * 4;
    if (!host_dst) return (int64_t)need;
    if (cap < need) return fail(h, "ev_get_stage(%s): need %zu bytes, cap %zu", name, need, cap);
    float* d_tmp = nullptr; int64_t* d_scr = nullptr;
    HIPCHK(h, hipMalloc((void**)&d_tmp, need));
    HIPCHK(h, hipMalloc((void**)&d_scr, (3 * (size_t)h->B + 8) * 8));
    int rc = pack_level(h, t.ptr, t.dtype, t.ld, t.C, t.shift, t.level == 0, d_tmp, d_scr);
    if (rc == 0 && hipStreamSynchronize(h->stream) != hipSuccess) rc = fail(h, "ev_get_stage: gather failed");
    if (rc == 0 && hipMemcpy(host_dst, d_tmp, need, hipMemcpyDeviceToHost) != hipSuccess) rc = fail(h, "ev_get_stage: D2H failed");
    (void)hipFree(d_tmp); (void)hipFree(d_scr);
    return rc ? -1 : (int64_t)need;
}

// ------------------------------------------------------------------- per-kernel test entry points (include/evhip_ops.h)
int ev_op_conv_gemm(const ev_conv_gemm_desc* d, void* stream) {
    static_assert(sizeof(ev_conv_gemm_desc) == sizeof(ConvGemmParams), "descriptor layout must match ConvGemmParams");
    ConvGemmParams p;
    memcpy(&p, d, sizeof p);
    const int es = p.dtype == DT_F16 ? 2 : 4;
    if (p.M % ROW_ALIGN || p.N % 32 || (p.K * es) % 64 || (p.taps - 1) * p.dil > 64) return -2;
    if (p.dtype == DT_F32S && (p.K % 32 || !p.W_lo)) return -2;
    if (p.dtype == DT_MX && (p.K % 32 || !p.W)) return -2;
    if (mx_check(p) || splitk_check(p)) return -2;
    if (!p.out16 && !p.out32 && !p.mxo_h) return -2;
    if (p.pro_lrelu && !(p.pro_slope >= 0.f && p.pro_slope <= 1.f)) return -2;
    launch_conv_gemm(p, (hipStream_t)stream);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}
size_t ev_op_mx_scratch_bytes(int M, int K) { return mx_scratch_bytes(M, K); }
int ev_op_resblock_pair_c32(const ev_res_pair_desc* d, void* stream) {
    static_assert(sizeof(ev_res_pair_desc) == sizeof(ResPairParams), "descriptor layout must match ResPairParams");
    ResPairParams p;
    memcpy(&p, d, sizeof p);
    if (p.k != 3 && p.k != 7 && p.k != 11) return -2;
    if (p.epi.post_lrelu && !(p.epi.post_slope >= 0.f && p.epi.post_slope <= 1.f)) return -2;   // max(v, s v) form of leaky-relu
    launch_resblock_pair_c32(p, (hipStream_t)stream);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}
int ev_op_resblock_pair_c32_mx(const ev_res_pair_desc* d, void* stream) {
    static_assert(sizeof(ev_res_pair_desc) == sizeof(ResPairParams), "descriptor layout must match ResPairParams");
    ResPairParams p;
    memcpy(&p, d, sizeof p);
    if (p.M <= 0 || p.dil < 1 || (p.k - 1) * p.dil > 64) return -2;
    if (launch_resblock_pair_c32_mx(p, (hipStream_t)stream)) return -2;
    return hipGetLastError() == hipSuccess ? 0 : -1;
}
int ev_op_resblock_pair_c64_mx(const ev_res_pair_desc* d, void* stream) {
    ResPairParams p;
    memcpy(&p, d, sizeof p);
    if (launch_resblock_pair_c64_mx(p, (hipStream_t)stream)) return -2;
    return hipGetLastError() == hipSuccess ? 0 : -1;
}
int ev_op_resblock_pair_c64(const ev_res_pair_desc* d, void* stream) {
    ResPairParams p;
    memcpy(&p, d, sizeof p);
    if (p.k != 3) return -2;
    if (p.epi.post_lrelu && !(p.epi.post_slope >= 0.f && p.epi.post_slope <= 1.f)) return -2;
    launch_resblock_pair_c64(p, (hipStream_t)stream);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}
int ev_op_layernorm(const float* x, int rows, int C, const float* gamma, const float* beta, float eps, const uint8_t* row_valid,
                    void* out16, float* out32, const float* dot_w, float dot_b, float* dot_out, void* stream) {
    LayerNormParams p{};
    p.x = x; p.ldx = C; p.rows = rows; p.C = C; p.gamma = gamma; p.beta = beta; p.eps = eps; p.row_valid = row_valid; p.out16 = out16;
    p.out32 = out32; p.ldo = C; p.dot_w = dot_w; p.dot_b = dot_b; p.dot_out = dot_out;
    launch_layernorm(p, (hipStream_t)stream);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}
int ev_op_layernorm_planes(const float* x, int rows, int C, const float* gamma, const float* beta, float eps, const uint8_t* row_valid,
                           void* h, void* q4h, void* q4l, void* qsh, void* qsl, unsigned qs_stride, void* stream) {
    if (C > 512 || C % 128 || !h || !q4h || !q4l || !qsh || !qsl) return -2;
    LayerNormParams p{};
    p.x = x; p.ldx = C; p.rows = rows; p.C = C; p.gamma = gamma; p.beta = beta; p.eps = eps; p.row_valid = row_valid; p.ldo = C;
    p.mxo_h = h; p.mxo_q4[0] = q4h; p.mxo_q4[1] = q4l; p.mxo_qs[0] = qsh; p.mxo_qs[1] = qsl; p.mxo_qs_stride = qs_stride;
    launch_layernorm(p, (hipStream_t)stream);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}
int ev_op_attention(const void* qkv, int is_f16, int C, int heads, const int32_t* seq_off, const int32_t* seq_len, int B, int max_len,
                    void* out, void* stream) {
    AttnParams p{};
    // is_f16 == 2: fp32 rows, split-precision products
    p.qkv = qkv; p.dtype = is_f16 == 1 ? DT_F16 : (is_f16 == 2 ? DT_F32S : DT_F32); p.ld = 3 * C; p.C = C; p.heads = heads; p.seq_off = seq_off; p.seq_len = seq_len;
    p.B = B; p.max_len = max_len; p.out = out; p.ldo = C;
    launch_attention(p, (hipStream_t)stream);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

}  // extern "C"
