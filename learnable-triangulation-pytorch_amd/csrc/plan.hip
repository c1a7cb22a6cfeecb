// Plan-level entry points of the C ABI ("whole-network plan entry points that own pre-packed weights and a hipGraph"):
//   lt_plan_create_vol / lt_plan_forward_vol / lt_plan_info / lt_plan_destroy
// A non-Python host hands over the reference's state_dict (names + host fp32 arrays) and the model configuration once, and then calls the forward with
// device images and host camera parameters: layer -> kernel selection, every batch threshold, the eval-BatchNorm fold, weight packing (GEMM layout,
// MFMA fragment orders, parity phases of the transposed convolutions, split-K tap groups), buffer reuse and the captured hipGraph all live behind this
// file -- the same rules lt_engine.py / mvn/models/*.py apply when the Python modules record their plan (tests/test_gpu_plan_abi.py holds the two
// against each other and against the reference's golden outputs).
//
// What it replaces in the reference (file:line): VolumetricTriangulationNet.__init__ / forward (mvn/models/triangulation.py:204-355), PoseResNet
// (mvn/models/pose_resnet.py:57-318: Bottleneck / BasicBlock / Bottleneck_CAFFE, _make_layer, _make_deconv_layer, GlobalAveragePoolingHead), V2VModel
// (mvn/models/v2v.py:7-180), Camera.update_after_resize / projection (mvn/utils/multiview.py:33-52), the cuboid of triangulation.py:281-341.
//
// Host-only code (no kernel here): every launch goes through the kernel-level entry points of include/lt_hip.h.
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include <deque>
#include <functional>
#include <map>
#include <string>
#include <unordered_map>
#include <vector>

#include "lt_common.h"

using namespace lt;

namespace {

#define PL_TRY(call)                   \
    do {                               \
        const int rc_ = (call);        \
        if (rc_ != LT_OK) return rc_;  \
    } while (0)
#define PL_HIP(call)                                                        \
    do {                                                                    \
        hipError_t e_ = (call);                                             \
        if (e_ != hipSuccess) {                                             \
            set_error("%s failed: %s", #call, hipGetErrorString(e_));       \
            return LT_ERR_LAUNCH;                                           \
        }                                                                   \
    } while (0)

constexpr float BN_EPS = 1e-5f;
constexpr int GEO_RING = 4;

bool envset(const char* k) { const char* e = getenv(k); return e && e[0] && !(e[0] == '0' && !e[1]); }

unsigned short bf16_rne(float f) {          // torch's float -> bfloat16 (round to nearest even, NaN kept quiet)
    unsigned u;
    memcpy(&u, &f, 4);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (unsigned short)((u >> 16) | 0x40);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (unsigned short)(u >> 16);
}

int cout_pad_of(int cout) { return cout <= 16 ? 16 : cout <= 32 ? 32 : cout <= 64 ? 64 : (cout + 127) / 128 * 128; }

// ---- host views of the caller's tensors --------------------------------------------------------------------------------------------
struct WT {                                   // a weight tensor of the state dict, or a host-made one (own)
    const float* d = nullptr;
    int nd = 0;
    int64_t s[5] = {0, 0, 0, 0, 0};
    std::vector<float> own;
    int64_t numel() const { int64_t n = 1; for (int i = 0; i < nd; ++i) n *= s[i]; return n; }
};
struct BN { const float* g = nullptr; const float* b = nullptr; const float* m = nullptr; const float* v = nullptr; };

// a channels-last activation on the device: [n][d][h][w][c], element size es
struct Act {
    char* p = nullptr;
    int n = 0, d = 0, h = 0, w = 0, c = 0, es = 0;
    bool pooled = true, planar = false;
    size_t bytes() const { return (size_t)n * d * h * w * c * es; }
    long long numel() const { return (long long)n * d * h * w * c; }
    bool null() const { return p == nullptr; }
};

struct PhaseSpec { std::vector<float> w; std::vector<int32_t> taps; int ntaps = 0; int off[3] = {0, 0, 0}; };
struct ConvSpec {
    int N, D, H, W, Cin, Do, Ho, Wo, st[3], pd[3], OD, OH, OW, ostr[3], Cout, cout_pad, k_pad, flags;
    std::vector<float> bias, scale, shift;
    std::vector<PhaseSpec> ph;
};

// Epilogue constants of y = (acc + bias) * scale + shift: eval-mode BatchNorm folded exactly the way ATen evaluates it (fp32: invstd = 1 / sqrt(var + eps),
// scale = invstd * weight, shift = bias_bn - mean * scale) -- lt_engine.fold_bn.  (The library is built with -ffp-contract=off: every step rounds like torch's.)
void fold_bn(int cout, const float* bias, const BN* bn, int cp, std::vector<float>& bi, std::vector<float>& sc, std::vector<float>& sh) {
    bi.assign(cp, 0.f); sc.assign(cp, 1.f); sh.assign(cp, 0.f);
    if (bias) for (int i = 0; i < cout; ++i) bi[i] = bias[i];
    if (bn)
        for (int i = 0; i < cout; ++i) {
            const float ve = bn->v[i] + BN_EPS;
            const float invstd = 1.0f / sqrtf(ve);
            const float alpha = invstd * bn->g[i];
            const float ma = bn->m[i] * alpha;
            sc[i] = alpha;
            sh[i] = bn->b[i] - ma;
        }
}

int floordiv(int a, int b) { int q = a / b, r = a % b; return (r != 0 && ((r < 0) != (b < 0))) ? q - 1 : q; }
int floormod(int a, int b) { return a - floordiv(a, b) * b; }

// lt_engine.make_conv_spec: the lt_conv_fwd description of one (transposed) convolution layer.  w: Conv{2,3}d [Cout][Cin][k..] or ConvTranspose{2,3}d
// [Cin][Cout][k..] (stride 2 only); in: (N, D, H, W, Cin_buffer) of the channels-last input (Cin_buffer >= Cin: extra input channels get zero weights).
int make_conv_spec(const WT& w, const float* bias, const BN* bn, const int in[5], int stride, int pad, int es, bool transposed, int flags, int output_padding, ConvSpec& sp) {
    const bool nd3 = w.nd == 5;
    LT_REQUIRE(w.nd == 4 || w.nd == 5, LT_ERR_INVALID, "plan: convolution weight with %d dimensions", w.nd);
    const int kd = nd3 ? (int)w.s[2] : 1, kh = (int)w.s[nd3 ? 3 : 2], kw = (int)w.s[nd3 ? 4 : 3];
    const int N = in[0], D = in[1], Hh = in[2], W = in[3], cin_buf = in[4];
    int st[3] = {stride, stride, stride}, pd[3] = {pad, pad, pad};
    if (!nd3) { st[0] = 1; pd[0] = 0; }
    const int kstep = 128 / es;
    sp.N = N; sp.D = D; sp.H = Hh; sp.W = W; sp.flags = flags;
    if (!transposed) {
        const int cout = (int)w.s[0], cin = (int)w.s[1];
        LT_REQUIRE(cin <= cin_buf, LT_ERR_INVALID, "plan: weight has %d input channels, the activation %d", cin, cin_buf);
        const int Do = (D + 2 * pd[0] - kd) / st[0] + 1, Ho = (Hh + 2 * pd[1] - kh) / st[1] + 1, Wo = (W + 2 * pd[2] - kw) / st[2] + 1;
        const int cp = cout_pad_of(cout), K = kd * kh * kw * cin_buf, k_pad = (K + kstep - 1) / kstep * kstep;
        sp.Cin = cin_buf; sp.Do = Do; sp.Ho = Ho; sp.Wo = Wo; sp.OD = Do; sp.OH = Ho; sp.OW = Wo;
        for (int i = 0; i < 3; ++i) { sp.st[i] = st[i]; sp.pd[i] = pd[i]; sp.ostr[i] = 1; }
        sp.Cout = cout; sp.cout_pad = cp; sp.k_pad = k_pad;
        fold_bn(cout, bias, bn, cp, sp.bias, sp.scale, sp.shift);
        sp.ph.resize(1);
        PhaseSpec& ph = sp.ph[0];
        ph.w.assign((size_t)cp * k_pad, 0.f);
        ph.ntaps = kd * kh * kw;
        const int64_t ktot = (int64_t)kd * kh * kw;
        for (int co = 0; co < cout; ++co)
            for (int ci = 0; ci < cin; ++ci) {
                const float* src = w.d + ((int64_t)co * cin + ci) * ktot;
                for (int t = 0; t < ktot; ++t) ph.w[(size_t)co * k_pad + (size_t)t * cin_buf + ci] = src[t];
            }
        for (int a = 0; a < kd; ++a)
            for (int b = 0; b < kh; ++b)
                for (int c = 0; c < kw; ++c) { ph.taps.push_back(a); ph.taps.push_back(b); ph.taps.push_back(c); ph.taps.push_back(((a * Hh + b) * W + c) * cin_buf); }
        return LT_OK;
    }
    // ---- stride-2 transposed convolution: one phase per output parity
    const int cin = (int)w.s[0], cout = (int)w.s[1];
    LT_REQUIRE(cin == cin_buf, LT_ERR_INVALID, "plan: the input of a transposed convolution may not be channel padded");
    LT_REQUIRE(stride == 2, LT_ERR_UNSUPPORTED, "plan: only stride-2 transposed convolutions");
    const int ks[3] = {kd, kh, kw}, dims[3] = {D, Hh, W};
    int outs[3];
    for (int i = 0; i < 3; ++i) {
        if (!nd3 && i == 0) { outs[i] = 1; continue; }
        const int o = (dims[i] - 1) * 2 - 2 * pd[i] + ks[i] + output_padding;
        LT_REQUIRE(o == 2 * dims[i], LT_ERR_UNSUPPORTED, "plan: a transposed convolution must exactly double the size (k = 4, p = 1 / k = 2, p = 0)");
        outs[i] = o;
    }
    const int cp = cout_pad_of(cout);
    struct DimPhase { int phi; std::vector<std::pair<int, int>> taps; };
    auto dim_phases = [&](int i) {
        std::vector<DimPhase> res;
        if (!nd3 && i == 0) { res.push_back({0, {{0, 0}}}); return res; }
        for (int phi = 0; phi < 2; ++phi) {          // o = 2 q + phi = 2 i_in - p + kk  ->  kk = phi + p (mod 2), i_in = q + (phi + p - kk) / 2
            DimPhase dp; dp.phi = phi;
            for (int kk = 0; kk < ks[i]; ++kk)
                if (floormod(phi + pd[i] - kk, 2) == 0) dp.taps.push_back({kk, floordiv(phi + pd[i] - kk, 2)});
            res.push_back(dp);
        }
        return res;
    };
    const int64_t ktot = (int64_t)kd * kh * kw;
    struct Raw { std::vector<std::pair<std::vector<float>, int>> dummy; };
    std::vector<PhaseSpec> phases;
    std::vector<std::vector<int>> phase_k;          // per phase: the filter tap index (ka, kb, kc flattened) of each recorded tap, -1 = zero tap
    int ntaps_max = 0;
    for (const DimPhase& A : dim_phases(0))
        for (const DimPhase& Bp : dim_phases(1))
            for (const DimPhase& Cp : dim_phases(2)) {
                PhaseSpec ph; std::vector<int> kidx;
                for (auto& ta : A.taps)
                    for (auto& tb : Bp.taps)
                        for (auto& tc : Cp.taps) {
                            const int da = ta.second, db = tb.second, dc = tc.second;
                            ph.taps.push_back(da); ph.taps.push_back(db); ph.taps.push_back(dc); ph.taps.push_back(((da * Hh + db) * W + dc) * cin);
                            kidx.push_back((ta.first * kh + tb.first) * kw + tc.first);
                        }
                if (kidx.empty()) { ph.taps = {0, 0, 0, 0}; kidx.push_back(-1); }          // an output parity no tap reaches: one tap with zero weights writes the zeros
                ph.ntaps = (int)kidx.size();
                ph.off[0] = A.phi; ph.off[1] = Bp.phi; ph.off[2] = Cp.phi;
                if (ph.ntaps > ntaps_max) ntaps_max = ph.ntaps;
                phases.push_back(ph); phase_k.push_back(kidx);
            }
    const int K = ntaps_max * cin, k_pad = (K + kstep - 1) / kstep * kstep;
    for (size_t p = 0; p < phases.size(); ++p) {
        PhaseSpec& ph = phases[p];
        ph.w.assign((size_t)cp * k_pad, 0.f);
        for (int t = 0; t < ph.ntaps; ++t) {
            const int kt = phase_k[p][t];
            if (kt < 0) continue;
            for (int ci = 0; ci < cin; ++ci)
                for (int co = 0; co < cout; ++co) ph.w[(size_t)co * k_pad + (size_t)t * cin + ci] = w.d[((int64_t)ci * cout + co) * ktot + kt];
        }
    }
    LT_REQUIRE((int)phases.size() <= LT_CONV_MAX_PHASES, LT_ERR_UNSUPPORTED, "plan: %d phases", (int)phases.size());
    sp.Cin = cin; sp.Do = D; sp.Ho = Hh; sp.Wo = W; sp.OD = outs[0]; sp.OH = outs[1]; sp.OW = outs[2];
    for (int i = 0; i < 3; ++i) { sp.st[i] = 1; sp.pd[i] = 0; }
    sp.ostr[0] = nd3 ? 2 : 1; sp.ostr[1] = 2; sp.ostr[2] = 2;
    sp.Cout = cout; sp.cout_pad = cp; sp.k_pad = k_pad;
    fold_bn(cout, bias, bn, cp, sp.bias, sp.scale, sp.shift);
    sp.ph = std::move(phases);
    return LT_OK;
}

}  // namespace

// =====================================================================================================================================
struct lt_plan {
    lt_vol_plan_config cfg;
    int dtype = LT_F32, es = 4;                 // element type / size of activations and weights
    std::unordered_map<std::string, const lt_named_tensor*> sd;
    std::vector<void*> allocs;                  // every hipMalloc of the plan
    std::map<std::pair<size_t, int>, std::vector<char*>> pool;          // released activations by (bytes, element size)
    size_t bytes_alloc = 0;
    double flops = 0;
    typedef std::function<int(hipStream_t)> Op;
    std::vector<Op> ops;                        // ops[0 .. npre) read the caller's images, ops[npre .. nops - ntail) are captured, the tail writes the caller's outputs
    int npre = 0, ntail = 0;
    std::deque<lt_conv_desc> conv_descs;        // descriptor storage with stable addresses (the launch closures hold pointers)
    std::deque<lt_conv_skip> skip_descs;
    std::deque<lt_conv_cat2> cat2_descs;
    std::deque<lt_pwchain_desc> pw_descs;
    std::deque<lt_stem_desc> stem_descs;
    std::deque<lt_bneck_desc> bneck_descs;
    std::deque<lt_bneck_ds_desc> bneck_ds_descs;
    std::deque<lt_xr_desc> xr_descs;
    hipGraphExec_t graph = nullptr;
    bool captured = false;
    hipStream_t own_stream = nullptr; hipEvent_t own_ev[2] = {nullptr, nullptr};          // stream == NULL with use_graph: the legacy default stream cannot be captured
    // per-call pointers the pre / tail ops read
    const float* cur_images = nullptr;
    float* out_kp = nullptr; float* out_probs = nullptr; float* out_feats = nullptr;
    // geometry block (fp32): proj B*NV*12 | pos B*3 | center B*3 | rot B*9 -- one H2D copy per forward from a ring of pinned blocks
    float* geo_dev = nullptr; float* geo_host[GEO_RING] = {nullptr, nullptr, nullptr, nullptr}; hipEvent_t geo_ev[GEO_RING] = {nullptr, nullptr, nullptr, nullptr};
    int geo_slot = 0; size_t n_geo = 0, o_pos = 0, o_cen = 0, o_rot = 0;
    // results
    Act feats, vol, logits, volc;
    float* coords = nullptr; float* kp = nullptr; float* probs = nullptr; void* sa_ws = nullptr;
    int hm_h = 0, hm_w = 0;
    int n_xr = 0, n_bneck = 0, n_bneck_ds = 0, n_cat2 = 0, n_halo2d = 0, n_pwchain = 0, n_stem = 0, n_splitk = 0, n_conv_skip = 0;

    ~lt_plan() {
        if (graph) (void)hipGraphExecDestroy(graph);
        if (own_stream) (void)hipStreamDestroy(own_stream);
        for (int i = 0; i < 2; ++i) if (own_ev[i]) (void)hipEventDestroy(own_ev[i]);
        for (void* p : allocs) (void)hipFree(p);
        for (int i = 0; i < GEO_RING; ++i) { if (geo_host[i]) (void)hipHostFree(geo_host[i]); if (geo_ev[i]) (void)hipEventDestroy(geo_ev[i]); }
    }

    // ---- memory ------------------------------------------------------------------------------------------------------------------
    int dev_alloc(size_t bytes, void** out) {
        void* p = nullptr;
        PL_HIP(hipMalloc(&p, bytes ? bytes : 16));
        allocs.push_back(p);
        bytes_alloc += bytes;
        *out = p;
        return LT_OK;
    }
    int alloc(int n, int d, int h, int w, int c, int esz, Act& a) {          // PlanBuilder.alloc: size-keyed reuse of released activations
        a = Act(); a.n = n; a.d = d; a.h = h; a.w = w; a.c = c; a.es = esz;
        auto it = pool.find({a.bytes(), esz});
        if (it != pool.end() && !it->second.empty()) { a.p = it->second.back(); it->second.pop_back(); return LT_OK; }
        void* p; PL_TRY(dev_alloc(a.bytes(), &p)); a.p = (char*)p;
        return LT_OK;
    }
    void release(const Act& a) { if (a.p && a.pooled) pool[{a.bytes(), a.es}].push_back(a.p); }          // stream order makes the reuse safe
    int upload(const void* host, size_t bytes, void** out) {
        PL_TRY(dev_alloc(bytes, out));
        PL_HIP(hipMemcpy(*out, host, bytes, hipMemcpyHostToDevice));
        return LT_OK;
    }
    int upload_f32(const std::vector<float>& v, const float** out) { void* p; PL_TRY(upload(v.data(), v.size() * 4, &p)); *out = (const float*)p; return LT_OK; }
    int upload_w(const std::vector<float>& v, const void** out) {          // weights in the plan's element type (bf16: round to nearest even like torch's .to(bfloat16))
        if (es == 4) { void* p; PL_TRY(upload(v.data(), v.size() * 4, &p)); *out = p; return LT_OK; }
        std::vector<unsigned short> h(v.size());
        for (size_t i = 0; i < v.size(); ++i) h[i] = bf16_rne(v[i]);
        void* p; PL_TRY(upload(h.data(), h.size() * 2, &p)); *out = p;
        return LT_OK;
    }
    int upload_i32(const std::vector<int32_t>& v, const int32_t** out) { void* p; PL_TRY(upload(v.data(), v.size() * 4, &p)); *out = (const int32_t*)p; return LT_OK; }

    // ---- state dict ----------------------------------------------------------------------------------------------------------------
    int get(const std::string& name, WT& w, int nd_expect = 0) {
        auto it = sd.find(name);
        LT_REQUIRE(it != sd.end(), LT_ERR_INVALID, "lt_plan_create_vol: state dict has no '%s'", name.c_str());
        const lt_named_tensor* t = it->second;
        LT_REQUIRE(t->data && t->ndim >= 1 && t->ndim <= 5 && (!nd_expect || t->ndim == nd_expect), LT_ERR_INVALID, "lt_plan_create_vol: '%s' has %d dimensions", name.c_str(), t->ndim);
        w.d = t->data; w.nd = t->ndim;
        for (int i = 0; i < t->ndim; ++i) w.s[i] = t->shape[i];
        return LT_OK;
    }
    bool has(const std::string& name) const { return sd.find(name) != sd.end(); }
    int get_vec(const std::string& name, int n, const float** out) {
        WT w; PL_TRY(get(name, w));
        LT_REQUIRE(w.numel() == n, LT_ERR_INVALID, "lt_plan_create_vol: '%s' has %lld elements, expected %d", name.c_str(), (long long)w.numel(), n);
        *out = w.d;
        return LT_OK;
    }
    int get_bn(const std::string& prefix, int c, BN& bn) {
        PL_TRY(get_vec(prefix + ".weight", c, &bn.g)); PL_TRY(get_vec(prefix + ".bias", c, &bn.b));
        PL_TRY(get_vec(prefix + ".running_mean", c, &bn.m)); PL_TRY(get_vec(prefix + ".running_var", c, &bn.v));
        return LT_OK;
    }

    // ---- ops ------------------------------------------------------------------------------------------------------------------------
    struct ConvOpt {
        const float* bias = nullptr; const BN* bn = nullptr;
        int stride = 1, pad = 0; bool transposed = false, relu = false, relu_pre = false, out_f32 = false, sigmoid = false;
        const Act* residual = nullptr;
        // lt_conv_skip_fwd: the residual is a 1x1x1 convolution + BatchNorm of this 16-channel tensor, computed inside the launch
        const Act* skip_x = nullptr; const WT* skip_w = nullptr; const float* skip_bias = nullptr; const BN* skip_bn = nullptr;
    };

    // PlanBuilder.can_conv_skip
    bool can_conv_skip(const int xs[5], const WT& w, const Act& sx, const WT& sw) {
        if (dtype != LT_BF16 || envset("LT_NO_CONV_SKIP") || envset("LT_HALO_NO_COL") || envset("LT_HALO_NO_PERSIST") || envset("LT_CONV_NO_HALO")) return false;
        if (!(w.nd == 5 && w.s[0] == 32 && w.s[1] == 32 && w.s[2] == 3 && w.s[3] == 3 && w.s[4] == 3)) return false;
        if (!(sw.nd == 5 && sw.s[0] == 32 && sw.s[1] == 16 && sw.s[2] == 1 && sw.s[3] == 1 && sw.s[4] == 1)) return false;
        const int N = xs[0], D = xs[1], Hh = xs[2], W = xs[3], Cin = xs[4];
        if (Cin != 32 || sx.n != N || sx.d != D || sx.h != Hh || sx.w != W || sx.c != 16 || D % 4 || Hh % 8 || W % 8 || D / 4 < 2) return false;
        const int nc = lt_conv_chunk_samples(N, (long long)D * Hh * W * 32);
        if (nc < 1) return false;
        const int last = N - (N - 1) / nc * nc;
        for (int n : {nc, last}) {
            const long long nblk = (long long)n * (D / 4) * (Hh / 8) * (W / 8), cols = (long long)n * (Hh / 8) * (W / 8);
            if (!(nblk >= 1024 && nblk % 8 == 0 && cols % 8 == 0 && cols >= 256)) return false;
        }
        return true;
    }

    // PlanBuilder.splitk_slices: V2V's 3^3 128 -> 128 layers at the 8^3 / 4^3 / 2^3 levels as S tap-group phases + lt_splitk_reduce
    int splitk_slices(const ConvSpec& sp, const WT& w, const ConvOpt& o) {
        if (dtype != LT_BF16 || o.transposed || o.out_f32 || o.sigmoid || envset("LT_CONV_NO_SPLITK")) return 1;
        if (w.nd != 5 || w.s[2] != 3 || w.s[3] != 3 || w.s[4] != 3 || sp.st[0] != 1 || sp.st[1] != 1 || sp.st[2] != 1 || sp.pd[0] != 1 || sp.pd[1] != 1 || sp.pd[2] != 1) return 1;
        if (sp.Cin < 128 || sp.Cin % 64 || sp.Cout % 4 || sp.Cout != sp.cout_pad || sp.D * sp.H * sp.W > 512) return 1;
        const long long rows = (long long)sp.N * sp.Do * sp.Ho * sp.Wo;
        const int bm = rows >= 8192 ? 128 : 64;
        const long long tiles = ((rows + bm - 1) / bm) * (sp.cout_pad / bm);
        long long S = 256 / tiles;
        if (S > 8) S = 8;
        if (S < 1) S = 1;
        return (int)S;
    }

    void fill_desc(lt_conv_desc& d, const ConvSpec& sp) {
        memset(&d, 0, sizeof(d));
        d.dtype = dtype;
        d.N = sp.N; d.D = sp.D; d.H = sp.H; d.W = sp.W; d.Cin = sp.Cin; d.Do = sp.Do; d.Ho = sp.Ho; d.Wo = sp.Wo;
        for (int i = 0; i < 3; ++i) { d.stride[i] = sp.st[i]; d.pad[i] = sp.pd[i]; d.out_stride[i] = sp.ostr[i]; }
        d.OD = sp.OD; d.OH = sp.OH; d.OW = sp.OW;
        d.Cout = sp.Cout; d.ldc = sp.Cout; d.cout_pad = sp.cout_pad; d.k_pad = sp.k_pad;
        d.nphase = (int)sp.ph.size(); d.flags = sp.flags; d.tile = 0; d.stages = 0;
    }

    int pack_frag(const void* wdev, size_t elems, int which, int cout_pad, int k_pad, int cin, int ntaps, const void** out) {
        void* p; PL_TRY(dev_alloc(elems * 2, &p));
        int rc;
        if (which == 2) rc = lt_conv_pack_weights_t32(wdev, cout_pad, k_pad, cin, ntaps, p, nullptr);
        else if (which == 3) rc = lt_conv_pack_weights32(wdev, cout_pad, k_pad, p, nullptr);
        else rc = lt_conv_pack_weights(wdev, cout_pad, k_pad, p, nullptr);
        PL_TRY(rc);
        PL_HIP(hipStreamSynchronize(nullptr));
        *out = p;
        return LT_OK;
    }

    // PlanBuilder.conv
    int conv(const Act& x, const WT& w, const ConvOpt& o, Act& y) {
        const int flags = (o.relu ? LT_EPI_RELU_POST : 0) | (o.relu_pre ? LT_EPI_RELU_PRE : 0) | (o.out_f32 ? LT_EPI_STORE_F32 : 0) | (o.sigmoid ? LT_EPI_SIGMOID : 0);
        const int in[5] = {x.n, x.d, x.h, x.w, x.c};
        ConvSpec sp;
        PL_TRY(make_conv_spec(w, o.bias, o.bn, in, o.stride, o.pad, es, o.transposed, flags, 0, sp));
        lt_conv_skip* sk = nullptr;
        if (o.skip_x) {
            LT_REQUIRE(!o.residual && !o.relu_pre && !o.out_f32 && can_conv_skip(in, w, *o.skip_x, *o.skip_w), LT_ERR_INVALID, "plan: conv_skip on an unsupported shape");
            const int sin[5] = {o.skip_x->n, o.skip_x->d, o.skip_x->h, o.skip_x->w, o.skip_x->c};
            ConvSpec ss;
            PL_TRY(make_conv_spec(*o.skip_w, o.skip_bias, o.skip_bn, sin, 1, 0, es, false, 0, 0, ss));
            LT_REQUIRE(ss.cout_pad == 32 && sp.cout_pad == 32, LT_ERR_INVALID, "plan: conv_skip widths");
            // the skip branch's BatchNorm: scale into its weights (fp32 product, ONE bf16 rounding), (bias * scale + shift) into this convolution's shift
            std::vector<float> wf((size_t)32 * ss.k_pad);
            for (int co = 0; co < 32; ++co) for (int k = 0; k < ss.k_pad; ++k) wf[(size_t)co * ss.k_pad + k] = ss.ph[0].w[(size_t)co * ss.k_pad + k] * ss.scale[co];
            for (int co = 0; co < 32; ++co) { const float bs = ss.bias[co] * ss.scale[co]; const float t = sp.shift[co] + bs; sp.shift[co] = t + ss.shift[co]; }          // Python's order: (shift + bias * scale) + shift_skip
            const void* wsk; PL_TRY(upload_w(wf, &wsk));
            const void* wfr; PL_TRY(pack_frag(wsk, 32 * 16, 2, 32, ss.k_pad, 16, 1, &wfr));
            skip_descs.emplace_back();
            sk = &skip_descs.back();
            sk->x = o.skip_x->p; sk->cin = 16; sk->weight_frag = wfr;
            ++n_conv_skip;
        }
        const int S = splitk_slices(sp, w, o);
        if (S > 1) return conv_splitk(x, w, sp, S, o.residual, y);
        PL_TRY(alloc(sp.N, sp.OD, sp.OH, sp.OW, sp.Cout, o.out_f32 ? 4 : es, y));
        if (o.residual) LT_REQUIRE(o.residual->n == y.n && o.residual->d == y.d && o.residual->h == y.h && o.residual->w == y.w && o.residual->c == y.c && o.residual->es == es,
                                   LT_ERR_INVALID, "plan: residual shape");
        conv_descs.emplace_back();
        lt_conv_desc& d = conv_descs.back();
        fill_desc(d, sp);
        const bool bf = dtype == LT_BF16;
        const bool k1 = [&] { for (int i = 2; i < w.nd; ++i) if (w.s[i] != 1) return false; return true; }();
        for (size_t i = 0; i < sp.ph.size(); ++i) {
            const PhaseSpec& ph = sp.ph[i];
            const void* wdev; PL_TRY(upload_w(ph.w, &wdev));
            const int32_t* tdev; PL_TRY(upload_i32(ph.taps, &tdev));
            d.phase[i].weight = wdev; d.phase[i].taps = tdev; d.phase[i].ntaps = ph.ntaps;
            for (int k = 0; k < 3; ++k) d.phase[i].out_off[k] = ph.off[k];
            const size_t elems = (size_t)sp.cout_pad * sp.k_pad;
            // ---- the weights ALSO in the MFMA fragment order of the kernel that will run the layer (lt_engine.PlanBuilder.conv's rules, bf16 plans)
            const bool w2d_3x3 = !o.transposed && w.nd == 4 && w.s[0] == 256 && w.s[1] == 256 && w.s[2] == 3 && w.s[3] == 3;
            const bool w2d_4x4t = o.transposed && w.nd == 4 && w.s[0] == 256 && w.s[1] == 256 && w.s[2] == 4 && w.s[3] == 4;
            bool all4 = true;
            for (auto& q : sp.ph) all4 = all4 && q.ntaps == 4;
            if (bf && x.c == 256 && sp.D == 1 && sp.W % 24 == 0 && sp.H % 8 == 0 && !o.residual && !o.out_f32 && !o.sigmoid && !envset("LT_CONV_NO_H2D") &&
                sp.Cout == 256 && sp.cout_pad == 256 && d.ldc % 8 == 0 && !envset("LT_CONV_V1") &&
                ((long long)sp.N * (sp.H / 8) * (sp.W / 24) >= 60 || envset("LT_H2D_ANY_SIZE")) &&
                ((w2d_3x3 && sp.st[1] == 1 && sp.st[2] == 1 && sp.pd[1] == 1 && sp.pd[2] == 1 && sp.W == 24 && sp.OH == sp.H && sp.OW == sp.W) ||
                 (w2d_4x4t && sp.ph.size() == 4 && !envset("LT_DECONV_NO_H2D") && sp.ostr[1] == 2 && sp.ostr[2] == 2 && sp.OH == 2 * sp.H && sp.OW == 2 * sp.W && all4))) {
                // ResNet layer3's 3x3 256 -> 256 on 24-wide maps and the head's 4x4 / stride-2 transposed 256 -> 256, from 60 tiles of 8 x 24 pixels on: conv2d_halo_kernel
                const void* wfr; PL_TRY(pack_frag(wdev, elems, 2, sp.cout_pad, sp.k_pad, 256, ph.ntaps, &wfr));
                d.phase[i].weight_frag = wfr; d.phase[i].weight_frag_layout = 2;
                if (i == 0) ++n_halo2d;
            } else if (bf && sp.cout_pad % 256 == 0 && sp.k_pad % 64 == 0) {
                // the 288-row layers: fragment order of the 32x32x16 MFMA (conv_igemm7); short-K pointwise layers: the 144-row variant of conv_igemm6 (16x16x32 order)
                const bool short_pw = k1 && sp.k_pad <= 256 && !o.transposed;
                const int layout = (!envset("LT_CONV_NO_V7") && !short_pw) ? 3 : 1;
                const void* wfr; PL_TRY(pack_frag(wdev, elems, layout, sp.cout_pad, sp.k_pad, 0, 0, &wfr));
                d.phase[i].weight_frag = wfr; d.phase[i].weight_frag_layout = layout;
            } else if (bf && !o.transposed && w.nd == 5 && x.c == w.s[1] && w.s[2] == 3 && w.s[3] == 3 && w.s[4] == 3 && sp.st[0] == 1 && sp.st[1] == 1 && sp.st[2] == 1 &&
                       sp.pd[0] == 1 && sp.pd[1] == 1 && sp.pd[2] == 1 &&
                       ((w.s[0] == 64 && w.s[1] == 64) || (w.s[0] == 64 && w.s[1] == 32) || (w.s[0] == 128 && w.s[1] == 128) || (w.s[0] == 32 && w.s[1] == 16))) {
                // V2V's 3x3x3 64 -> 64, 32 -> 64, 128 -> 128, 16 -> 32: fragments of the transposed product for conv3d_halo_wreg_kernel
                const void* wfr; PL_TRY(pack_frag(wdev, elems, 2, sp.cout_pad, sp.k_pad, (int)w.s[1], 27, &wfr));
                d.phase[i].weight_frag = wfr; d.phase[i].weight_frag_layout = 2;
            }
        }
        const float *bi, *sc, *sh;
        PL_TRY(upload_f32(sp.bias, &bi)); PL_TRY(upload_f32(sp.scale, &sc)); PL_TRY(upload_f32(sp.shift, &sh));
        long long taps = 0;
        for (auto& q : sp.ph) taps += q.ntaps;
        flops += 2.0 * sp.N * sp.Do * sp.Ho * sp.Wo * sp.Cout * taps * (o.transposed ? w.s[0] : w.s[1]);
        const lt_conv_desc* dp = &d;
        const void* xp = x.p; void* yp = y.p; const void* rp = o.residual ? o.residual->p : nullptr;
        if (sk) {
            flops += 2.0 * sp.N * sp.Do * sp.Ho * sp.Wo * sp.Cout * 16;
            ops.push_back([=](hipStream_t s) { return lt_conv_skip_fwd(dp, xp, bi, sc, sh, sk, yp, s); });
        } else
            ops.push_back([=](hipStream_t s) { return lt_conv_fwd(dp, xp, bi, sc, sh, rp, yp, s); });
        return LT_OK;
    }

    // PlanBuilder._conv_splitk
    int conv_splitk(const Act& x, const WT& w, const ConvSpec& sp, int S, const Act* residual, Act& y) {
        const PhaseSpec& ph0 = sp.ph[0];
        const int ntaps = ph0.ntaps, Cin = sp.Cin, kstep = 128 / es;
        std::vector<int> bounds(S + 1);
        int maxg = 0;
        for (int i = 0; i <= S; ++i) bounds[i] = ntaps * i / S;
        for (int i = 0; i < S; ++i) if (bounds[i + 1] - bounds[i] > maxg) maxg = bounds[i + 1] - bounds[i];
        const int kp = (maxg * Cin + kstep - 1) / kstep * kstep;
        Act part; PL_TRY(alloc(sp.N, S * sp.Do, sp.Ho, sp.Wo, sp.Cout, 4, part));
        PL_TRY(alloc(sp.N, sp.Do, sp.Ho, sp.Wo, sp.Cout, es, y));
        const long long rows = (long long)sp.N * sp.Do * sp.Ho * sp.Wo;
        conv_descs.emplace_back();
        lt_conv_desc& d = conv_descs.back();
        fill_desc(d, sp);
        d.OD = S * sp.Do; d.k_pad = kp; d.nphase = S; d.flags = LT_EPI_STORE_F32;
        d.tile = rows >= 8192 ? LT_TILE2_128x128 : LT_TILE2_64x64;
        for (int i = 0; i < S; ++i) {
            const int t0 = bounds[i], t1 = bounds[i + 1];
            std::vector<float> wk((size_t)sp.cout_pad * kp, 0.f);
            for (int co = 0; co < sp.cout_pad; ++co)
                for (int k = t0 * Cin; k < t1 * Cin; ++k) wk[(size_t)co * kp + (k - t0 * Cin)] = ph0.w[(size_t)co * sp.k_pad + k];
            std::vector<int32_t> tp(ph0.taps.begin() + 4 * t0, ph0.taps.begin() + 4 * t1);
            const void* wdev; PL_TRY(upload_w(wk, &wdev));
            const int32_t* tdev; PL_TRY(upload_i32(tp, &tdev));
            d.phase[i].weight = wdev; d.phase[i].taps = tdev; d.phase[i].ntaps = t1 - t0;
            d.phase[i].out_off[0] = i * sp.Do; d.phase[i].out_off[1] = 0; d.phase[i].out_off[2] = 0;
        }
        std::vector<float> zero(sp.cout_pad, 0.f), one(sp.cout_pad, 1.f);
        const float *ibi, *isc, *ish, *bi, *sc, *sh;
        PL_TRY(upload_f32(zero, &ibi)); PL_TRY(upload_f32(one, &isc)); PL_TRY(upload_f32(zero, &ish));
        PL_TRY(upload_f32(sp.bias, &bi)); PL_TRY(upload_f32(sp.scale, &sc)); PL_TRY(upload_f32(sp.shift, &sh));
        flops += 2.0 * rows * sp.Cout * ntaps * w.s[1];
        const lt_conv_desc* dp = &d;
        const void* xp = x.p; float* pp = (float*)part.p; void* yp = y.p; const void* rp = residual ? residual->p : nullptr;
        ops.push_back([=](hipStream_t s) { return lt_conv_fwd(dp, xp, ibi, isc, ish, nullptr, pp, s); });
        const int dt = dtype, N = sp.N, C = sp.Cout, fl = sp.flags;
        const long long R = (long long)sp.Do * sp.Ho * sp.Wo;
        ops.push_back([=](hipStream_t s) { return lt_splitk_reduce(dt, pp, S, N, R, C, bi, sc, sh, rp, yp, fl, s); });
        release(part);
        ++n_splitk;
        return LT_OK;
    }

    // PlanBuilder.can_conv_cat2 / conv_cat2: expand + (strided) downsample branch of a Bottleneck's first block as ONE pointwise convolution over [t2 | x]
    bool can_conv_cat2(const int t2s[5], const WT& we, const Act& x, const WT& wd, int sds) {
        if (dtype != LT_BF16 || envset("LT_NO_CONV_CAT2") || envset("LT_CONV_NO_V7") || envset("LT_CONV_NO_V3")) return false;
        const int N = t2s[0], D = t2s[1], Ho = t2s[2], Wo = t2s[3], P = t2s[4];
        if (D != 1 || we.nd != 4 || wd.nd != 4 || we.s[2] != 1 || we.s[3] != 1 || wd.s[2] != 1 || wd.s[3] != 1) return false;
        const int Cc = (int)we.s[0], Cin2 = (int)wd.s[1];
        if (we.s[1] != P || wd.s[0] != Cc || (sds != 1 && sds != 2) || x.n != N || x.d != 1 || x.h != Ho * sds || x.w != Wo * sds || x.c != Cin2) return false;
        if (P % 32 || Cin2 % 32 || (P + Cin2) % 64 || Cc % 256 || (P & (P - 1))) return false;
        const long long tiles = (((long long)N * Ho * Wo + 287) / 288) * (Cc / 256);
        if (tiles < 200 && !envset("LT_CAT2_ANY_SIZE")) return false;
        return (long long)N * Ho * Wo * Cc < (1ll << 31) && (long long)N * x.h * x.w * Cin2 < (1ll << 31);
    }
    int conv_cat2(const Act& t2, const WT& we, const BN& bne, const Act& x, const WT& wd, const BN& bnd, int sds, Act& y) {
        const int N = t2.n, Ho = t2.h, Wo = t2.w, P = t2.c, Cc = (int)we.s[0], Cin2 = (int)wd.s[1];
        const int t2s[5] = {t2.n, t2.d, t2.h, t2.w, t2.c}, xs[5] = {x.n, x.d, x.h, x.w, x.c};
        ConvSpec s3, sd_;
        PL_TRY(make_conv_spec(we, nullptr, &bne, t2s, 1, 0, es, false, LT_EPI_RELU_POST, 0, s3));
        PL_TRY(make_conv_spec(wd, nullptr, &bnd, xs, sds, 0, es, false, 0, 0, sd_));
        LT_REQUIRE(s3.cout_pad == Cc && sd_.cout_pad == Cc, LT_ERR_INVALID, "plan: conv_cat2 widths");
        const int kp = P + Cin2;
        std::vector<float> wcat((size_t)Cc * kp), shift(Cc), zero(Cc, 0.f);
        for (int co = 0; co < Cc; ++co) {
            for (int k = 0; k < P; ++k) wcat[(size_t)co * kp + k] = s3.ph[0].w[(size_t)co * s3.k_pad + k] * s3.scale[co];
            for (int k = 0; k < Cin2; ++k) wcat[(size_t)co * kp + P + k] = sd_.ph[0].w[(size_t)co * sd_.k_pad + k] * sd_.scale[co];
            const float a = s3.bias[co] * s3.scale[co], a2 = a + s3.shift[co], b = sd_.bias[co] * sd_.scale[co], b2 = b + sd_.shift[co];
            shift[co] = a2 + b2;
        }
        PL_TRY(alloc(N, 1, Ho, Wo, Cc, es, y));
        conv_descs.emplace_back();
        lt_conv_desc& d = conv_descs.back();
        memset(&d, 0, sizeof(d));
        d.dtype = dtype; d.N = N; d.D = 1; d.H = Ho; d.W = Wo; d.Cin = P; d.Do = 1; d.Ho = Ho; d.Wo = Wo;
        for (int i = 0; i < 3; ++i) { d.stride[i] = 1; d.pad[i] = 0; d.out_stride[i] = 1; }
        d.OD = 1; d.OH = Ho; d.OW = Wo; d.Cout = Cc; d.ldc = Cc; d.cout_pad = Cc; d.k_pad = kp; d.nphase = 1; d.flags = LT_EPI_RELU_POST;
        const void* wdev; PL_TRY(upload_w(wcat, &wdev));
        const int32_t* tdev; PL_TRY(upload_i32(s3.ph[0].taps, &tdev));
        const void* wfr; PL_TRY(pack_frag(wdev, (size_t)Cc * kp, 3, Cc, kp, 0, 0, &wfr));
        d.phase[0].weight = wdev; d.phase[0].taps = tdev; d.phase[0].ntaps = 1; d.phase[0].weight_frag = wfr; d.phase[0].weight_frag_layout = 3;
        cat2_descs.emplace_back();
        lt_conv_cat2& c2 = cat2_descs.back();
        c2.x = x.p; c2.cin = Cin2; c2.H = x.h; c2.W = x.w; c2.stride = sds;
        const float *bi, *sh;
        PL_TRY(upload_f32(zero, &bi)); PL_TRY(upload_f32(shift, &sh));
        flops += 2.0 * N * Ho * Wo * Cc * kp;
        const lt_conv_desc* dp = &d; const lt_conv_cat2* cp = &c2;
        const void* xp = t2.p; void* yp = y.p;
        ops.push_back([=](hipStream_t s) { return lt_conv_cat2_fwd(dp, xp, cp, bi, nullptr, sh, nullptr, yp, s); });
        ++n_cat2;
        return LT_OK;
    }

    // PlanBuilder.can_chain_pointwise / pwchain: V2V's pointwise tail in one pass, planar fp32 logits
    struct PwLayer { WT w; const float* bias; const BN* bn; bool relu; };
    bool can_chain_pointwise(const Act& x, const std::vector<PwLayer>& L) {
        if (dtype != LT_BF16 || L.empty() || (int)L.size() > LT_PWCHAIN_MAX || x.c != 32) return false;
        if (((long long)x.n * x.d * x.h * x.w) % 64) return false;
        int cin = 32;
        for (size_t i = 0; i < L.size(); ++i) {
            const WT& w = L[i].w;
            for (int k = 2; k < w.nd; ++k) if (w.s[k] != 1) return false;
            if (w.s[1] != cin || w.s[0] > 32) return false;
            if (i + 1 < L.size() && w.s[0] != 32) return false;
            cin = (int)w.s[0];
        }
        return true;
    }
    int pwchain(const Act& x, const std::vector<PwLayer>& L, Act& y) {
        pw_descs.emplace_back();
        lt_pwchain_desc& d = pw_descs.back();
        memset(&d, 0, sizeof(d));
        d.dtype = dtype; d.nlayers = (int)L.size(); d.rows = (long long)x.n * x.d * x.h * x.w; d.cin = 32;
        int shape[5] = {x.n, x.d, x.h, x.w, x.c};
        int cout_last = 0;
        for (size_t i = 0; i < L.size(); ++i) {
            const bool last = i + 1 == L.size();
            const int fl = (L[i].relu ? LT_EPI_RELU_POST : 0) | (last ? LT_EPI_STORE_F32 : 0);
            ConvSpec sp; PL_TRY(make_conv_spec(L[i].w, L[i].bias, L[i].bn, shape, 1, 0, es, false, fl, 0, sp));
            const void* wdev; PL_TRY(upload_w(sp.ph[0].w, &wdev));
            const float *bi, *sc, *sh;
            PL_TRY(upload_f32(sp.bias, &bi)); PL_TRY(upload_f32(sp.scale, &sc)); PL_TRY(upload_f32(sp.shift, &sh));
            d.cout[i] = sp.Cout; d.k_pad[i] = sp.k_pad; d.flags[i] = fl; d.weight[i] = wdev; d.bias[i] = bi; d.scale[i] = sc; d.shift[i] = sh;
            flops += 2.0 * d.rows * sp.Cout * L[i].w.s[1];
            shape[4] = sp.Cout; cout_last = sp.Cout;
        }
        d.ldy = cout_last;
        const long long vox = (long long)x.d * x.h * x.w;
        const bool planar = vox % 64 == 0;
        if (planar) d.plane = vox;
        PL_TRY(alloc(x.n, x.d, x.h, x.w, cout_last, 4, y));
        y.planar = planar; y.pooled = !planar;
        const lt_pwchain_desc* dp = &d; const void* xp = x.p; void* yp = y.p;
        ops.push_back([=](hipStream_t s) { return lt_pwchain_fwd(dp, xp, yp, s); });
        ++n_pwchain;
        return LT_OK;
    }

    int maxpool(const Act& x, int k, int s, int p, int nd, Act& y) {
        int32_t kk[3] = {nd == 2 ? 1 : k, k, k}, ss[3] = {nd == 2 ? 1 : s, s, s}, pp[3] = {nd == 2 ? 0 : p, p, p};
        const int od = (x.d + 2 * pp[0] - kk[0]) / ss[0] + 1, oh = (x.h + 2 * pp[1] - kk[1]) / ss[1] + 1, ow = (x.w + 2 * pp[2] - kk[2]) / ss[2] + 1;
        PL_TRY(alloc(x.n, od, oh, ow, x.c, x.es, y));
        const int dt = dtype; const Act xa = x; void* yp = y.p;
        std::vector<int32_t> a = {kk[0], kk[1], kk[2], ss[0], ss[1], ss[2], pp[0], pp[1], pp[2]};
        ops.push_back([=](hipStream_t st) { return lt_maxpool_fwd(dt, xa.p, yp, xa.n, xa.d, xa.h, xa.w, xa.c, a.data(), a.data() + 3, a.data() + 6, st); });
        return LT_OK;
    }

    int global_avgpool(const Act& x, Act& y) {          // [N,1,H,W,C] -> [1,1,1,N,C]: a one-row "image" of N pixels, feeds 1x1 convolutions = linears
        PL_TRY(alloc(1, 1, 1, x.n, x.c, x.es, y));
        const int dt = dtype; const Act xa = x; void* yp = y.p;
        ops.push_back([=](hipStream_t st) { return lt_global_avgpool(dt, xa.p, yp, xa.n, xa.d * xa.h * xa.w, xa.c, st); });
        return LT_OK;
    }

    // ---- whole-block launches of the bf16 plan (lt_engine.PlanBuilder.bottleneck / bottleneck_ds / expand_reduce / stem_pool) ------------------------
    int pack_block_layer(const WT& w, const BN& bn, const int in[5], int pad, int fl, const void** wfr, const float** sc, const float** sh, ConvSpec& sp) {
        PL_TRY(make_conv_spec(w, nullptr, &bn, in, 1, pad, es, false, fl, 0, sp));
        LT_REQUIRE(sp.cout_pad == sp.Cout && sp.k_pad == sp.ph[0].ntaps * sp.Cin, LT_ERR_INVALID, "plan: fused block layer padding (%d / %d)", sp.cout_pad, sp.k_pad);
        const void* wdev; PL_TRY(upload_w(sp.ph[0].w, &wdev));
        PL_TRY(pack_frag(wdev, (size_t)sp.cout_pad * sp.k_pad, 2, sp.cout_pad, sp.k_pad, sp.Cin, sp.ph[0].ntaps, wfr));
        PL_TRY(upload_f32(sp.scale, sc)); PL_TRY(upload_f32(sp.shift, sh));
        flops += 2.0 * in[0] * in[2] * in[3] * sp.Cout * sp.ph[0].ntaps * sp.Cin;
        return LT_OK;
    }
    bool can_bottleneck(const Act& x, const WT w[3], const int strides[3]) {
        if (dtype != LT_BF16 || envset("LT_NO_BNECK")) return false;
        if (x.d != 1 || strides[0] != 1 || strides[1] != 1 || strides[2] != 1) return false;
        const int Cc = x.c, P = (int)w[0].s[0];
        if (!((Cc == 256 && P == 64) || (Cc == 512 && P == 128))) return false;
        if (!(w[0].s[0] == P && w[0].s[1] == Cc && w[0].s[2] == 1 && w[1].s[0] == P && w[1].s[1] == P && w[1].s[2] == 3 && w[1].s[3] == 3 && w[2].s[0] == Cc && w[2].s[1] == P && w[2].s[2] == 1)) return false;
        return x.h % 8 == 0 && x.w % 16 == 0 && (long long)x.n * x.h * x.w * Cc < (1ll << 31);
    }
    int bottleneck(const Act& x, const WT w[3], const BN bn[3], Act& y) {
        PL_TRY(alloc(x.n, 1, x.h, x.w, x.c, es, y));
        bneck_descs.emplace_back();
        lt_bneck_desc& d = bneck_descs.back();
        memset(&d, 0, sizeof(d));
        d.dtype = dtype; d.N = x.n; d.H = x.h; d.W = x.w; d.C = x.c; d.P = (int)w[0].s[0];
        int shape[5] = {x.n, 1, x.h, x.w, x.c};
        for (int i = 0; i < 3; ++i) {
            ConvSpec sp;
            PL_TRY(pack_block_layer(w[i], bn[i], shape, i == 1 ? 1 : 0, LT_EPI_RELU_POST, &d.weight[i], &d.scale[i], &d.shift[i], sp));
            shape[4] = sp.Cout;
        }
        const lt_bneck_desc* dp = &d; const void* xp = x.p; void* yp = y.p;
        ops.push_back([=](hipStream_t s) { return lt_bottleneck_fwd(dp, xp, yp, s); });
        ++n_bneck;
        return LT_OK;
    }
    bool can_bottleneck_ds(const Act& x, const WT w[3], const int strides[3], const WT& wd, int sds) {
        if (dtype != LT_BF16 || envset("LT_NO_BNECK") || envset("LT_NO_BNECK_DS")) return false;
        if (x.d != 1 || strides[0] != 1 || strides[1] != 1 || strides[2] != 1 || sds != 1) return false;
        const int P = (int)w[0].s[0], Cc = (int)w[2].s[0];
        if (!(x.c == 64 && P == 64 && Cc == 256)) return false;
        if (!(w[0].s[1] == 64 && w[0].s[2] == 1 && w[1].s[0] == P && w[1].s[1] == P && w[1].s[2] == 3 && w[1].s[3] == 3 && w[2].s[1] == P && w[2].s[2] == 1 &&
              wd.s[0] == Cc && wd.s[1] == 64 && wd.s[2] == 1 && wd.s[3] == 1)) return false;
        return x.h % 8 == 0 && x.w % 16 == 0 && (long long)x.n * x.h * x.w * Cc < (1ll << 31);
    }
    int bottleneck_ds(const Act& x, const WT w[3], const BN bn[3], const WT& wd, const BN& bnd, Act& y) {
        const int Cc = (int)w[2].s[0];
        PL_TRY(alloc(x.n, 1, x.h, x.w, Cc, es, y));
        bneck_ds_descs.emplace_back();
        lt_bneck_ds_desc& d = bneck_ds_descs.back();
        memset(&d, 0, sizeof(d));
        d.dtype = dtype; d.N = x.n; d.H = x.h; d.W = x.w; d.Cin = x.c; d.P = (int)w[0].s[0]; d.C = Cc;
        int shape[5] = {x.n, 1, x.h, x.w, x.c};
        for (int i = 0; i < 3; ++i) {
            ConvSpec sp;
            PL_TRY(pack_block_layer(w[i], bn[i], shape, i == 1 ? 1 : 0, LT_EPI_RELU_POST, &d.weight[i], &d.scale[i], &d.shift[i], sp));
            shape[4] = sp.Cout;
        }
        const int xs[5] = {x.n, 1, x.h, x.w, x.c};
        ConvSpec sp;
        PL_TRY(pack_block_layer(wd, bnd, xs, 0, 0, &d.weight[3], &d.scale[3], &d.shift[3], sp));
        const lt_bneck_ds_desc* dp = &d; const void* xp = x.p; void* yp = y.p;
        ops.push_back([=](hipStream_t s) { return lt_bottleneck_ds_fwd(dp, xp, yp, s); });
        ++n_bneck_ds;
        return LT_OK;
    }
    bool can_expand_reduce(const Act& t2, const Act& res, const WT& we, const WT& wr) {
        if (dtype != LT_BF16 || envset("LT_NO_XR")) return false;
        if (t2.d != 1 || res.d != 1 || t2.n != res.n || t2.h != res.h || t2.w != res.w) return false;
        if (!(res.c == 1024 && t2.c == 256)) return false;
        if ((long long)t2.n * t2.h * t2.w < 36 * 96 && !envset("LT_XR_ANY_SIZE")) return false;
        return we.s[0] == 1024 && we.s[1] == 256 && we.s[2] == 1 && we.s[3] == 1 && wr.s[0] == 256 && wr.s[1] == 1024 && wr.s[2] == 1 && wr.s[3] == 1;
    }
    int expand_reduce(const Act& t2, const Act& res, const WT& we, const BN& bne, const WT& wr, const BN& bnr, Act& y, Act& t1) {
        const int N = t2.n, Hh = t2.h, W = t2.w, P = t2.c, Cc = res.c;
        PL_TRY(alloc(N, 1, Hh, W, Cc, es, y));
        PL_TRY(alloc(N, 1, Hh, W, P, es, t1));
        xr_descs.emplace_back();
        lt_xr_desc& d = xr_descs.back();
        memset(&d, 0, sizeof(d));
        d.dtype = dtype; d.C = Cc; d.P = P; d.M = (long long)N * Hh * W;
        const int s2[5] = {t2.n, 1, Hh, W, P}, sr[5] = {res.n, 1, Hh, W, Cc};
        ConvSpec s3, s1;
        PL_TRY(pack_block_layer(we, bne, s2, 0, LT_EPI_RELU_POST, &d.weight[0], &d.scale[0], &d.shift[0], s3));
        PL_TRY(pack_block_layer(wr, bnr, sr, 0, LT_EPI_RELU_POST, &d.weight[1], &d.scale[1], &d.shift[1], s1));
        std::vector<float> packed;          // the four tables back to back: one LDS-DMA in the kernel's prologue
        packed.insert(packed.end(), s3.scale.begin(), s3.scale.end()); packed.insert(packed.end(), s3.shift.begin(), s3.shift.end());
        packed.insert(packed.end(), s1.scale.begin(), s1.scale.end()); packed.insert(packed.end(), s1.shift.begin(), s1.shift.end());
        PL_TRY(upload_f32(packed, &d.consts));
        const lt_xr_desc* dp = &d; const void* a = t2.p; const void* r = res.p; void* yp = y.p; void* tp = t1.p;
        ops.push_back([=](hipStream_t s) { return lt_expand_reduce_fwd(dp, a, r, yp, tp, s); });
        ++n_xr;
        return LT_OK;
    }
    // conv1 + bn1 + relu + maxpool in one pass, reading the caller's fp32 (N,3,H,W) images: the first op of the plan, outside the captured graph
    int stem_pool(int N, int Hh, int W, const WT& w, const BN& bn, Act& y) {
        const int in[5] = {N, 1, Hh, W, 8};
        ConvSpec sp;
        PL_TRY(make_conv_spec(w, nullptr, &bn, in, 2, 3, es, false, LT_EPI_RELU_POST, 0, sp));
        const int Hp = (sp.Ho - 1) / 2 + 1, Wp = (sp.Wo - 1) / 2 + 1;
        PL_TRY(alloc(N, 1, Hp, Wp, 64, es, y));
        const void* wdev; PL_TRY(upload_w(sp.ph[0].w, &wdev));
        void* wpk; PL_TRY(dev_alloc(lt_stem_packed_bytes(), &wpk));
        PL_TRY(lt_stem_pack_weights(wdev, sp.k_pad, wpk, nullptr));
        PL_HIP(hipStreamSynchronize(nullptr));
        stem_descs.emplace_back();
        lt_stem_desc& d = stem_descs.back();
        memset(&d, 0, sizeof(d));
        d.dtype = dtype; d.N = N; d.H = Hh; d.W = W; d.Cin = 3; d.Cout = 64; d.weight = wpk; d.x_layout = 1;
        PL_TRY(upload_f32(sp.bias, &d.bias)); PL_TRY(upload_f32(sp.scale, &d.scale)); PL_TRY(upload_f32(sp.shift, &d.shift));
        flops += 2.0 * N * sp.Ho * sp.Wo * 64 * 49 * w.s[1];
        const lt_stem_desc* dp = &d; void* yp = y.p; lt_plan* self = this;
        LT_REQUIRE(ops.empty(), LT_ERR_INVALID, "plan: an op reading the caller's tensor must be the first of the plan");
        ops.push_back([=](hipStream_t s) { return lt_stem_pool_fwd(dp, self->cur_images, yp, s); });
        npre = 1;
        ++n_stem;
        return LT_OK;
    }

    // ---- the networks ---------------------------------------------------------------------------------------------------------------
    struct Block {          // one residual block of the backbone (pose_resnet.py:57-137)
        bool bottleneck; int nconv; WT w[3]; BN bn[3]; int strides[3]; bool has_ds; WT wd; BN bnd; int sds;
        bool identity_bottleneck() const { return bottleneck && !has_ds && strides[0] == 1 && strides[1] == 1 && strides[2] == 1; }
    };
    int load_block(const std::string& pre, bool bott, bool caffe, int stride, bool ds, Block& b) {
        b.bottleneck = bott; b.nconv = bott ? 3 : 2; b.has_ds = ds; b.sds = stride;
        if (bott) { b.strides[0] = caffe ? stride : 1; b.strides[1] = caffe ? 1 : stride; b.strides[2] = 1; }
        else { b.strides[0] = stride; b.strides[1] = 1; b.strides[2] = 1; }
        for (int i = 0; i < b.nconv; ++i) {
            PL_TRY(get(pre + ".conv" + std::to_string(i + 1) + ".weight", b.w[i], 4));
            PL_TRY(get_bn(pre + ".bn" + std::to_string(i + 1), (int)b.w[i].s[0], b.bn[i]));
        }
        if (ds) { PL_TRY(get(pre + ".downsample.0.weight", b.wd, 4)); PL_TRY(get_bn(pre + ".downsample.1", (int)b.wd.s[0], b.bnd)); }
        return LT_OK;
    }
    // ResidualBlock.record.  t1: this block's first activation when the previous block's seam launch produced it; nxt: fuse this block's expand with that
    // block's reduce where the plan supports the shape (t1n is then the next block's first activation, null otherwise).
    int record_block(const Block& b, const Act& x, const Act* t1_in, const Block* nxt, Act& y, Act& t1n) {
        t1n = Act();
        if (b.bottleneck && !b.has_ds && !t1_in && !nxt && can_bottleneck(x, b.w, b.strides)) return bottleneck(x, b.w, b.bn, y);
        if (b.bottleneck && b.has_ds && !t1_in && !nxt && can_bottleneck_ds(x, b.w, b.strides, b.wd, b.sds)) return bottleneck_ds(x, b.w, b.bn, b.wd, b.bnd, y);
        if (t1_in || nxt) {
            Act t1, t2;
            ConvOpt o;
            if (t1_in) t1 = *t1_in;
            else { o = ConvOpt(); o.bn = &b.bn[0]; o.relu = true; PL_TRY(conv(x, b.w[0], o, t1)); }
            o = ConvOpt(); o.bn = &b.bn[1]; o.relu = true; o.pad = 1; PL_TRY(conv(t1, b.w[1], o, t2));
            release(t1);
            if (nxt && can_expand_reduce(t2, x, b.w[2], nxt->w[0])) {
                PL_TRY(expand_reduce(t2, x, b.w[2], b.bn[2], nxt->w[0], nxt->bn[0], y, t1n));
                release(t2);
                return LT_OK;
            }
            o = ConvOpt(); o.bn = &b.bn[2]; o.relu = true; o.residual = &x; PL_TRY(conv(t2, b.w[2], o, y));
            release(t2);
            return LT_OK;
        }
        if (b.bottleneck && b.has_ds) {
            const int Ho = (x.h - 1) / b.sds + 1, Wo = (x.w - 1) / b.sds + 1;
            const int t2s[5] = {x.n, 1, Ho, Wo, (int)b.w[2].s[1]};
            if (can_conv_cat2(t2s, b.w[2], x, b.wd, b.sds)) {
                Act t1, t2; ConvOpt o;
                o.bn = &b.bn[0]; o.relu = true; o.stride = b.strides[0]; PL_TRY(conv(x, b.w[0], o, t1));
                o = ConvOpt(); o.bn = &b.bn[1]; o.relu = true; o.stride = b.strides[1]; o.pad = 1; PL_TRY(conv(t1, b.w[1], o, t2));
                release(t1);
                PL_TRY(conv_cat2(t2, b.w[2], b.bn[2], x, b.wd, b.bnd, b.sds, y));
                release(t2);
                return LT_OK;
            }
        }
        Act res = x; bool own_res = false;
        if (b.has_ds) { ConvOpt o; o.bn = &b.bnd; o.stride = b.sds; PL_TRY(conv(x, b.wd, o, res)); own_res = true; }
        Act cur = x;
        for (int i = 0; i < b.nconv; ++i) {
            const bool last = i + 1 == b.nconv;
            ConvOpt o; o.bn = &b.bn[i]; o.relu = true; o.stride = b.strides[i]; o.pad = (int)b.w[i].s[2] / 2; o.residual = last ? &res : nullptr;
            Act z; PL_TRY(conv(cur, b.w[i], o, z));
            if (cur.p != x.p) release(cur);
            cur = z;
        }
        if (own_res) release(res);
        y = cur;
        return LT_OK;
    }

    // GlobalAveragePoolingHead.record (pose_resnet.py:140-174): conv + BN (+ ReLU, which commutes with the max pool) -> pool, twice; global average; three linears
    // as 1x1 convolutions over an N-pixel map; sigmoid.  Returns Act [1,1,1,N,n] fp32
    int record_gap_head(const std::string& pre, const Act& x, Act& out) {
        WT w0, w4; const float *b0, *b4; BN bn1, bn5;
        PL_TRY(get(pre + ".features.0.weight", w0, 4)); PL_TRY(get_vec(pre + ".features.0.bias", (int)w0.s[0], &b0)); PL_TRY(get_bn(pre + ".features.1", (int)w0.s[0], bn1));
        PL_TRY(get(pre + ".features.4.weight", w4, 4)); PL_TRY(get_vec(pre + ".features.4.bias", (int)w4.s[0], &b4)); PL_TRY(get_bn(pre + ".features.5", (int)w4.s[0], bn5));
        Act y, p, g;
        ConvOpt o; o.bias = b0; o.bn = &bn1; o.pad = 1; o.relu = true; PL_TRY(conv(x, w0, o, y));
        PL_TRY(maxpool(y, 2, 2, 0, 2, p)); release(y);
        o = ConvOpt(); o.bias = b4; o.bn = &bn5; o.pad = 1; o.relu = true; PL_TRY(conv(p, w4, o, y)); release(p);
        PL_TRY(maxpool(y, 2, 2, 0, 2, p)); release(y);
        PL_TRY(global_avgpool(p, g)); release(p);
        Act cur = g;
        for (int i = 0; i < 3; ++i) {
            WT lw; const float* lb;
            PL_TRY(get(pre + ".head." + std::to_string(2 * i) + ".weight", lw, 2));
            PL_TRY(get_vec(pre + ".head." + std::to_string(2 * i) + ".bias", (int)lw.s[0], &lb));
            lw.nd = 4; lw.s[2] = 1; lw.s[3] = 1;          // Linear [out][in] read as a 1x1 convolution
            ConvOpt lo; lo.bias = lb; lo.relu = i < 2; lo.sigmoid = i == 2; lo.out_f32 = i == 2;
            Act z; PL_TRY(conv(cur, lw, lo, z)); release(cur);
            cur = z;
        }
        out = cur;
        return LT_OK;
    }

    // PoseResNet.record without the (dead in the volumetric path) heatmap layer: stem, four stages, optional vol_confidences head, three 4x4 / stride-2 deconvolutions
    int record_backbone(Act& feats256, Act& volconf) {
        const int N = cfg.B * cfg.NV, Hh = cfg.H, W = cfg.W;
        int nb[4]; bool bott;
        switch (cfg.num_layers) {
            case 18: bott = false; nb[0] = 2; nb[1] = 2; nb[2] = 2; nb[3] = 2; break;
            case 34: bott = false; nb[0] = 3; nb[1] = 4; nb[2] = 6; nb[3] = 3; break;
            case 50: bott = true; nb[0] = 3; nb[1] = 4; nb[2] = 6; nb[3] = 3; break;
            case 101: bott = true; nb[0] = 3; nb[1] = 4; nb[2] = 23; nb[3] = 3; break;
            case 152: bott = true; nb[0] = 3; nb[1] = 8; nb[2] = 36; nb[3] = 3; break;
            default: set_error("lt_plan_create_vol: num_layers %d (18 / 34 / 50 / 101 / 152)", cfg.num_layers); return LT_ERR_INVALID;
        }
        const bool caffe = cfg.style_caffe != 0;
        if (caffe) bott = true;          // the reference swaps in Bottleneck_CAFFE (expansion 4) at EVERY depth (pose_resnet.py:322-324)
        const int exp = bott ? 4 : 1;
        WT w1; BN bn1;
        PL_TRY(get("backbone.conv1.weight", w1, 4)); PL_TRY(get_bn("backbone.bn1", 64, bn1));
        Act y;
        if (dtype == LT_BF16 && w1.s[0] == 64 && w1.s[1] == 3 && w1.s[2] == 7 && w1.s[3] == 7) PL_TRY(stem_pool(N, Hh, W, w1, bn1, y));
        else {
            // fp32 plans: the images go through lt_nchw_to_nhwc (pre op) into a 4-channel map, then conv1 + max pool
            const int cpad = 16 / es;
            Act xin; PL_TRY(alloc(N, 1, Hh, W, cpad, es, xin)); xin.pooled = false;
            const int dt = dtype; lt_plan* self = this; void* xp = xin.p;
            ops.push_back([=](hipStream_t s) { return lt_nchw_to_nhwc(dt, self->cur_images, xp, N, 3, Hh * W, cpad, s); });
            npre = 1;
            ConvOpt o; o.bn = &bn1; o.stride = 2; o.pad = 3; o.relu = true;
            Act c1; PL_TRY(conv(xin, w1, o, c1));
            PL_TRY(maxpool(c1, 3, 2, 1, 2, y)); release(c1);
        }
        int inplanes = 64;
        const int planes_of[4] = {64, 128, 256, 512};
        for (int li = 0; li < 4; ++li) {
            const int planes = planes_of[li], stride = li == 0 ? 1 : 2;
            std::vector<Block> blocks(nb[li]);
            for (int bi = 0; bi < nb[li]; ++bi) {
                const int st = bi == 0 ? stride : 1;
                const bool ds = bi == 0 && (st != 1 || inplanes != planes * exp);
                PL_TRY(load_block("backbone.layer" + std::to_string(li + 1) + "." + std::to_string(bi), bott, caffe, st, ds, blocks[bi]));
                inplanes = planes * exp;
            }
            Act t1; bool have_t1 = false;
            for (int bi = 0; bi < nb[li]; ++bi) {
                const Block& blk = blocks[bi];
                const Block* nxt = bi + 1 < nb[li] ? &blocks[bi + 1] : nullptr;
                // a run of identity bottleneck blocks whose seams lt_expand_reduce_fwd covers (ResNet layer3, bf16 plans): expand(i) + reduce(i + 1) in one launch
                const bool chain = nxt && blk.identity_bottleneck() && nxt->identity_bottleneck() && dtype == LT_BF16 && blk.w[2].s[0] == 1024 && blk.w[2].s[1] == 256 &&
                                   nxt->w[0].s[0] == 256 && nxt->w[0].s[1] == 1024;
                Act z, t1n;
                if (chain) PL_TRY(record_block(blk, y, have_t1 ? &t1 : nullptr, nxt, z, t1n));
                else if (have_t1) PL_TRY(record_block(blk, y, &t1, nullptr, z, t1n));
                else PL_TRY(record_block(blk, y, nullptr, nullptr, z, t1n));
                release(y);
                y = z; t1 = t1n; have_t1 = !t1n.null();
            }
        }
        volconf = Act();
        if (cfg.aggregation == LT_AGG_CONF || cfg.aggregation == LT_AGG_CONF_NORM) PL_TRY(record_gap_head("backbone.vol_confidences", y, volconf));
        for (int i = 0; i < 3; ++i) {
            WT dw; BN dbn;
            PL_TRY(get("backbone.deconv_layers." + std::to_string(3 * i) + ".weight", dw, 4));
            PL_TRY(get_bn("backbone.deconv_layers." + std::to_string(3 * i + 1), (int)dw.s[1], dbn));
            const float* db = nullptr;
            if (has("backbone.deconv_layers." + std::to_string(3 * i) + ".bias")) PL_TRY(get_vec("backbone.deconv_layers." + std::to_string(3 * i) + ".bias", (int)dw.s[1], &db));
            LT_REQUIRE(dw.s[2] == 4 && dw.s[3] == 4, LT_ERR_UNSUPPORTED, "lt_plan_create_vol: only the 4x4 stride-2 deconvolution of the reference configs");
            ConvOpt o; o.bias = db; o.bn = &dbn; o.stride = 2; o.pad = 1; o.transposed = true; o.relu = true;
            Act z; PL_TRY(conv(y, dw, o, z)); release(y);
            y = z;
        }
        feats256 = y;
        return LT_OK;
    }

    // V2V (mvn/models/v2v.py): Res3DBlock / Basic3DBlock / Pool3DBlock / Upsample3DBlock / EncoderDecorder / V2VModel.record
    struct C3 { WT w; const float* b = nullptr; BN bn; bool has_bn = false; };
    int load_c3(const std::string& conv, const std::string& bnname, bool transposed, C3& c) {
        PL_TRY(get(conv + ".weight", c.w, 5));
        const int cout = (int)(transposed ? c.w.s[1] : c.w.s[0]);
        PL_TRY(get_vec(conv + ".bias", cout, &c.b));
        if (!bnname.empty()) { PL_TRY(get_bn(bnname, cout, c.bn)); c.has_bn = true; }
        return LT_OK;
    }
    int res3d(const std::string& pre, const Act& x, Act& out) {
        C3 c0, c3, sk;
        PL_TRY(load_c3(pre + ".res_branch.0", pre + ".res_branch.1", false, c0));
        PL_TRY(load_c3(pre + ".res_branch.3", pre + ".res_branch.4", false, c3));
        const bool has_skip = has(pre + ".skip_con.0.weight");
        if (has_skip) PL_TRY(load_c3(pre + ".skip_con.0", pre + ".skip_con.1", false, sk));
        const int mid[5] = {x.n, x.d, x.h, x.w, (int)c0.w.s[0]};
        Act y, z;
        if (has_skip && can_conv_skip(mid, c3.w, x, sk.w)) {
            ConvOpt o; o.bias = c0.b; o.bn = &c0.bn; o.pad = 1; o.relu = true; PL_TRY(conv(x, c0.w, o, y));
            o = ConvOpt(); o.bias = c3.b; o.bn = &c3.bn; o.pad = 1; o.relu = true; o.skip_x = &x; o.skip_w = &sk.w; o.skip_bias = sk.b; o.skip_bn = &sk.bn;
            PL_TRY(conv(y, c3.w, o, z)); release(y);
            out = z;
            return LT_OK;
        }
        Act skip = x; bool own = false;
        if (has_skip) { ConvOpt o; o.bias = sk.b; o.bn = &sk.bn; PL_TRY(conv(x, sk.w, o, skip)); own = true; }
        ConvOpt o; o.bias = c0.b; o.bn = &c0.bn; o.pad = 1; o.relu = true; PL_TRY(conv(x, c0.w, o, y));
        o = ConvOpt(); o.bias = c3.b; o.bn = &c3.bn; o.pad = 1; o.relu = true; o.residual = &skip; PL_TRY(conv(y, c3.w, o, z));
        release(y);
        if (own) release(skip);
        out = z;
        return LT_OK;
    }
    int record_v2v(const Act& vol_in, Act& logits_out) {
        const std::string V = "volume_net.";
        Act x = vol_in, y;
        {   // front_layers: Basic3DBlock(32, 16, 7), Res3DBlock(16, 32), Res3DBlock(32, 32) x 2
            C3 c; PL_TRY(load_c3(V + "front_layers.0.block.0", V + "front_layers.0.block.1", false, c));
            ConvOpt o; o.bias = c.b; o.bn = &c.bn; o.pad = (int)(c.w.s[2] - 1) / 2; o.relu = true;
            PL_TRY(conv(x, c.w, o, y)); release(x); x = y;
            for (int i = 1; i <= 3; ++i) { PL_TRY(res3d(V + "front_layers." + std::to_string(i), x, y)); release(x); x = y; }
        }
        {   // EncoderDecorder.record
            const std::string E = V + "encoder_decoder.";
            Act skips[5];
            for (int lvl = 1; lvl <= 5; ++lvl) {
                PL_TRY(res3d(E + "skip_res" + std::to_string(lvl), x, skips[lvl - 1]));
                Act p; PL_TRY(maxpool(x, 2, 2, 0, 3, p)); release(x);
                PL_TRY(res3d(E + "encoder_res" + std::to_string(lvl), p, x)); release(p);
            }
            PL_TRY(res3d(E + "mid_res", x, y)); release(x); x = y;
            for (int lvl = 5; lvl >= 1; --lvl) {
                PL_TRY(res3d(E + "decoder_res" + std::to_string(lvl), x, y)); release(x);
                C3 u; PL_TRY(load_c3(E + "decoder_upsample" + std::to_string(lvl) + ".block.0", E + "decoder_upsample" + std::to_string(lvl) + ".block.1", true, u));
                ConvOpt o; o.bias = u.b; o.bn = &u.bn; o.stride = 2; o.pad = 0; o.transposed = true; o.relu_pre = true; o.residual = &skips[lvl - 1];
                PL_TRY(conv(y, u.w, o, x));          // relu(BN(deconv y)) + skip: the decoder's skip add rides in the same epilogue (v2v.py:121-136)
                release(y); release(skips[lvl - 1]);
            }
        }
        PL_TRY(res3d(V + "back_layers.0", x, y)); release(x); x = y;
        C3 t1, t2, ol;
        PL_TRY(load_c3(V + "back_layers.1.block.0", V + "back_layers.1.block.1", false, t1));
        PL_TRY(load_c3(V + "back_layers.2.block.0", V + "back_layers.2.block.1", false, t2));
        PL_TRY(load_c3(V + "output_layer", "", false, ol));
        std::vector<PwLayer> chain = {{t1.w, t1.b, &t1.bn, true}, {t2.w, t2.b, &t2.bn, true}, {ol.w, ol.b, nullptr, false}};
        if (can_chain_pointwise(x, chain)) {          // the pointwise tail (back_layers[1:] + output_layer) as ONE pass, (N, J, V, V, V) planar fp32 logits
            PL_TRY(pwchain(x, chain, logits_out)); release(x);
            return LT_OK;
        }
        for (C3* c : {&t1, &t2}) { ConvOpt o; o.bias = c->b; o.bn = &c->bn; o.relu = true; PL_TRY(conv(x, c->w, o, y)); release(x); x = y; }
        ConvOpt o; o.bias = ol.b; o.out_f32 = true;
        PL_TRY(conv(x, ol.w, o, logits_out)); release(x);
        return LT_OK;
    }

    // VolumetricTriangulationNet._build_plan
    int build() {
        const int B = cfg.B, NV = cfg.NV, V = cfg.volume_size, J = cfg.num_joints;
        Act f256;
        PL_TRY(record_backbone(f256, volc));
        WT pw; const float* pb;
        PL_TRY(get("process_features.0.weight", pw, 4)); PL_TRY(get_vec("process_features.0.bias", (int)pw.s[0], &pb));
        LT_REQUIRE(pw.s[0] == 32, LT_ERR_UNSUPPORTED, "lt_plan_create_vol: process_features has %d output channels (32)", (int)pw.s[0]);
        ConvOpt o; o.bias = pb;
        PL_TRY(conv(f256, pw, o, feats)); release(f256);
        feats.pooled = false;
        hm_h = feats.h; hm_w = feats.w;
        // geometry block + its pinned staging ring
        n_geo = (size_t)B * NV * 12 + (size_t)B * 15; o_pos = (size_t)B * NV * 12; o_cen = o_pos + 3 * B; o_rot = o_pos + 6 * B;
        void* g; PL_TRY(dev_alloc(n_geo * 4, &g)); geo_dev = (float*)g;
        for (int i = 0; i < GEO_RING; ++i) { PL_HIP(hipHostMalloc((void**)&geo_host[i], n_geo * 4, hipHostMallocDefault)); }
        void* c; PL_TRY(dev_alloc((size_t)B * V * V * V * 3 * 4, &c)); coords = (float*)c;
        PL_TRY(alloc(B, V, V, V, 32, es, vol));
        const float step = (float)(cfg.cuboid_side / (double)(V - 1));          // float(np.float32(side / (V - 1))): the fp64 quotient rounded once
        {
            const int dt = dtype, agg = cfg.aggregation, cmu = cfg.transfer_cmu_to_human36m ? 1 : 0, h = hm_h, w = hm_w;
            const float* gp = geo_dev; const size_t op = o_pos, oc = o_cen, orr = o_rot;
            const void* fp = feats.p; float* cp = coords; void* vp = vol.p; const float* confp = volc.null() ? nullptr : (const float*)volc.p;
            ops.push_back([=](hipStream_t s) { return lt_unproject_grid_fwd(dt, fp, gp, gp + op, gp + oc, gp + orr, step, cmu, cp, confp, vp, B, NV, 32, h, w, V, agg, s); });
        }
        PL_TRY(record_v2v(vol, logits));
        void* k; PL_TRY(dev_alloc((size_t)B * J * 3 * 4, &k)); kp = (float*)k;
        PL_TRY(dev_alloc((size_t)B * J * V * V * V * 4, &k)); probs = (float*)k;
        const long long nvox = (long long)V * V * V;
        const size_t wsb = lt_softargmax3d_workspace(B, J, nvox);
        PL_TRY(dev_alloc(wsb ? wsb : 16, &sa_ws));
        {   // tail op 1: the RETURNED features (B, NV, 32, h, w) fp32, skipped when the caller did not ask for them
            const int dt = dtype, h = hm_h, w = hm_w; const void* fp = feats.p; lt_plan* self = this;
            ops.push_back([=](hipStream_t s) { return self->out_feats ? lt_nhwc_to_nchw_f32(dt, fp, self->out_feats, B * NV, 32, h * w, 32, s) : LT_OK; });
        }
        {   // tail op 2: soft-argmax straight into the caller's tensors
            const float mult = (float)cfg.volume_multiplier; const int sm = cfg.volume_softmax ? 1 : 0, cl = logits.planar ? 0 : 1;
            const float* lp = (const float*)logits.p; const float* cp = coords; void* ws = sa_ws; lt_plan* self = this;
            float* kpd = kp; float* prd = probs;
            ops.push_back([=](hipStream_t s) {
                return lt_softargmax3d_fwd(lp, cp, mult, sm, cl, J, self->out_kp ? self->out_kp : kpd, self->out_probs ? self->out_probs : prd, B, J, nvox, ws, s);
            });
        }
        ntail = 2;
        return LT_OK;
    }

    int run(hipStream_t st) {
        const int nops = (int)ops.size();
        for (int i = 0; i < npre; ++i) PL_TRY(ops[i](st));
        if (cfg.use_graph && !captured) {
            // warm-up launch outside capture (sets function attributes, loads code objects), then capture the middle section once
            for (int i = npre; i < nops - ntail; ++i) PL_TRY(ops[i](st));
            PL_HIP(hipStreamSynchronize(st));
            PL_HIP(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
            int rc = LT_OK;
            for (int i = npre; i < nops - ntail && rc == LT_OK; ++i) rc = ops[i](st);
            hipGraph_t g = nullptr;
            const hipError_t e = hipStreamEndCapture(st, &g);
            if (rc != LT_OK) { if (g) (void)hipGraphDestroy(g); return rc; }
            if (e != hipSuccess) { set_error("hipStreamEndCapture: %s", hipGetErrorString(e)); return LT_ERR_LAUNCH; }
            const hipError_t e2 = hipGraphInstantiate(&graph, g, nullptr, nullptr, 0);
            (void)hipGraphDestroy(g);
            if (e2 != hipSuccess) { set_error("hipGraphInstantiate: %s", hipGetErrorString(e2)); return LT_ERR_LAUNCH; }
            captured = true;
        } else if (cfg.use_graph) {
            PL_HIP(hipGraphLaunch(graph, st));
        } else {
            for (int i = npre; i < nops - ntail; ++i) PL_TRY(ops[i](st));
        }
        for (int i = nops - ntail; i < nops; ++i) PL_TRY(ops[i](st));
        return LT_OK;
    }
};

extern "C" int lt_plan_create_vol(const lt_vol_plan_config* cfg, const lt_named_tensor* weights, int32_t nweights, lt_plan** plan_out) {
    LT_REQUIRE(cfg && weights && plan_out && nweights > 0, LT_ERR_INVALID, "lt_plan_create_vol: null argument");
    LT_REQUIRE(cfg->dtype == LT_F32 || cfg->dtype == LT_BF16, LT_ERR_INVALID, "lt_plan_create_vol: dtype %d", cfg->dtype);
    LT_REQUIRE(cfg->B >= 1 && cfg->NV >= 1 && cfg->H >= 32 && cfg->W >= 32 && cfg->volume_size >= 2 && cfg->num_joints >= 1, LT_ERR_INVALID, "lt_plan_create_vol: bad shape");
    LT_REQUIRE(cfg->aggregation >= LT_AGG_SUM && cfg->aggregation <= LT_AGG_CONF_NORM, LT_ERR_INVALID, "lt_plan_create_vol: aggregation %d", cfg->aggregation);
    *plan_out = nullptr;
    lt_plan* p = new lt_plan();
    p->cfg = *cfg;
    p->dtype = cfg->dtype; p->es = cfg->dtype == LT_F32 ? 4 : 2;
    for (int i = 0; i < nweights; ++i) {
        if (!weights[i].name) { delete p; set_error("lt_plan_create_vol: weight %d has no name", i); return LT_ERR_INVALID; }
        std::string n = weights[i].name;
        if (n.rfind("module.", 0) == 0) n = n.substr(7);          // DataParallel checkpoints (train.py:408-410)
        p->sd[n] = &weights[i];
    }
    const int rc = p->build();
    if (rc != LT_OK) { delete p; return rc; }
    p->sd.clear();          // the caller's host arrays are not referenced after this call
    *plan_out = p;
    return LT_OK;
}

extern "C" int lt_plan_forward_vol(lt_plan* p, const float* images, const double* K_host, const double* R_host, const double* t_host, const double* base_points_host,
                                   const double* rot_host, float* keypoints_3d, float* volumes, float* features, float* coord_volumes, float* vol_confidences,
                                   void* stream) {
    LT_REQUIRE(p && images && K_host && R_host && t_host && base_points_host && keypoints_3d, LT_ERR_INVALID, "lt_plan_forward_vol: null argument");
    hipStream_t st = (hipStream_t)stream;
    const bool detour = st == nullptr && p->cfg.use_graph;
    if (detour) {          // the plan's own stream, ordered behind what the default stream has queued so far
        if (!p->own_stream) {
            PL_HIP(hipStreamCreateWithFlags(&p->own_stream, hipStreamNonBlocking));
            for (int i = 0; i < 2; ++i) PL_HIP(hipEventCreateWithFlags(&p->own_ev[i], hipEventDisableTiming));
        }
        PL_HIP(hipEventRecord(p->own_ev[0], nullptr));
        PL_HIP(hipStreamWaitEvent(p->own_stream, p->own_ev[0], 0));
        st = p->own_stream;
    }
    const int B = p->cfg.B, NV = p->cfg.NV, V = p->cfg.volume_size;
    // ---- host geometry in fp64 like the reference (triangulation.py:272-296): Camera.update_after_resize to the heatmap resolution, projection = K [R | t];
    // cuboid position = base - side / 2; rotation about the vertical axis (identity in eval mode); one pinned block, one H2D copy
    const int slot = p->geo_slot = (p->geo_slot + 1) % GEO_RING;
    if (p->geo_ev[slot]) PL_HIP(hipEventSynchronize(p->geo_ev[slot]));          // the copy that last read this slot (GEO_RING forwards ago) has completed
    else PL_HIP(hipEventCreateWithFlags(&p->geo_ev[slot], hipEventDisableTiming));
    float* gh = p->geo_host[slot];
    const double sx = (double)p->hm_w / (double)p->cfg.W, sy = (double)p->hm_h / (double)p->cfg.H;
    for (int i = 0; i < B * NV; ++i) {
        double K[9];
        for (int k = 0; k < 9; ++k) K[k] = K_host[(size_t)i * 9 + k];
        K[0] *= sx; K[4] *= sy; K[2] *= sx; K[5] *= sy;
        const double* R = R_host + (size_t)i * 9; const double* t = t_host + (size_t)i * 3;
        for (int r = 0; r < 3; ++r)
            for (int c = 0; c < 4; ++c) {
                double acc = 0.0;
                for (int k = 0; k < 3; ++k) acc += K[r * 3 + k] * (c < 3 ? R[k * 3 + c] : t[k]);
                gh[(size_t)i * 12 + r * 4 + c] = (float)acc;
            }
    }
    const double half = (double)p->cfg.cuboid_side / 2.0;
    for (int b = 0; b < B; ++b) {
        for (int k = 0; k < 3; ++k) {
            gh[p->o_pos + 3 * b + k] = (float)(base_points_host[3 * b + k] - half);
            gh[p->o_cen + 3 * b + k] = (float)base_points_host[3 * b + k];
        }
        for (int k = 0; k < 9; ++k) gh[p->o_rot + 9 * b + k] = rot_host ? (float)rot_host[9 * b + k] : ((k == 0 || k == 4 || k == 8) ? 1.f : 0.f);
    }
    PL_HIP(hipMemcpyAsync(p->geo_dev, gh, p->n_geo * 4, hipMemcpyHostToDevice, st));
    PL_HIP(hipEventRecord(p->geo_ev[slot], st));
    p->cur_images = images; p->out_kp = keypoints_3d; p->out_probs = volumes; p->out_feats = features;
    PL_TRY(p->run(st));
    if (coord_volumes) PL_HIP(hipMemcpyAsync(coord_volumes, p->coords, (size_t)B * V * V * V * 3 * 4, hipMemcpyDeviceToDevice, st));
    if (vol_confidences) {
        LT_REQUIRE(!p->volc.null(), LT_ERR_INVALID, "lt_plan_forward_vol: vol_confidences asked of a plan without the confidence head (aggregation %d)", p->cfg.aggregation);
        PL_HIP(hipMemcpyAsync(vol_confidences, p->volc.p, (size_t)B * NV * 32 * 4, hipMemcpyDeviceToDevice, st));          // RAW sigmoid outputs; 'conf_norm' divides by their sum over views
    }
    if (detour) {          // ... and in front of what the default stream gets next
        PL_HIP(hipEventRecord(p->own_ev[1], st));
        PL_HIP(hipStreamWaitEvent(nullptr, p->own_ev[1], 0));
    }
    return LT_OK;
}

extern "C" int lt_plan_info(const lt_plan* p, lt_plan_info_t* info) {
    LT_REQUIRE(p && info, LT_ERR_INVALID, "lt_plan_info: null argument");
    memset(info, 0, sizeof(*info));
    info->launches = (int32_t)p->ops.size();
    info->heatmap_h = p->hm_h; info->heatmap_w = p->hm_w;
    info->flops = p->flops; info->bytes_allocated = (int64_t)p->bytes_alloc;
    info->n_expand_reduce = p->n_xr; info->n_bottleneck = p->n_bneck; info->n_bottleneck_ds = p->n_bneck_ds; info->n_conv_cat2 = p->n_cat2;
    info->n_conv2d_halo = p->n_halo2d; info->n_pwchain = p->n_pwchain; info->n_stem_pool = p->n_stem; info->n_splitk = p->n_splitk; info->n_conv_skip = p->n_conv_skip;
    info->graph_captured = p->captured ? 1 : 0;
    info->logits = (const float*)p->logits.p; info->logits_planar = p->logits.planar ? 1 : 0;
    return LT_OK;
}

extern "C" void lt_plan_destroy(lt_plan* p) { delete p; }
