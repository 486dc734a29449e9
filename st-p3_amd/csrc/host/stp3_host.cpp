// stp3_host.cpp -- EXPERIMENTAL (STP3_CPP_OPS=1) C++ launch path for the custom operators.
//
// The eager training step is host-bound (GPU busy ~55 %): every custom operator costs
// 60-140 us of Python per call (ctypes marshalling, a dozen small torch calls, autograd.Function glue).  This
// extension implements the same three operators -- fused BatchNorm + activation, bf16 MFMA convolution, depthwise
// convolution -- as torch::autograd::Function subclasses in C++ on top of the SAME C ABI (include/stp3_hip.h,
// libstp3hip.so is dlopen()ed, nothing is linked or duplicated), mirroring stp3_amd/ops.py line by line.
// Single-process only: with torch.distributed initialised the BatchNorm stays on the Python path (it owns the
// cross-replica all-reduce between the statistics and the apply pass).
//
// Built by csrc/Makefile with plain g++ against the torch headers (no device code here).
#include <torch/extension.h>
#include <ATen/ATen.h>
#include <c10/hip/HIPStream.h>
#include <cstdlib>
#include <dlfcn.h>

#include <string>
#include <unordered_map>

#include "stp3_hip.h"

namespace {

using at::Tensor;
using torch::autograd::AutogradContext;
using torch::autograd::variable_list;

// ---- the C ABI, resolved at init() ------------------------------------------------------------------------------
#define STP3_SYMBOLS(X)                                                                                              \
    X(stp3_bn_fwd_train) X(stp3_bn_bwd_train) X(stp3_bn_apply_fwd) X(stp3_bn_bwd_reduce) X(stp3_bn_apply_bwd)       \
    X(stp3_conv2d_fwd) X(stp3_conv2d_wgrad) X(stp3_conv2d_wgrad_workspace)                                            \
    X(stp3_dwconv2d_fwd) X(stp3_dwconv2d_bwd_data) X(stp3_dwconv2d_bwd_weight) X(stp3_dwconv2d_bwd_weight_workspace)

struct Api {
#define X(name) decltype(&::name) name = nullptr;
    STP3_SYMBOLS(X)
#undef X
} api;
bool g_ready = false;

void init(const std::string& lib_path) {
    void* h = dlopen(lib_path.c_str(), RTLD_NOW | RTLD_GLOBAL);
    TORCH_CHECK(h != nullptr, "stp3_host: cannot load ", lib_path, ": ", dlerror());
#define X(name)                                                        \
    api.name = reinterpret_cast<decltype(&::name)>(dlsym(h, #name)); \
    TORCH_CHECK(api.name != nullptr, "stp3_host: libstp3hip.so does not export " #name);
    STP3_SYMBOLS(X)
#undef X
    g_ready = true;
}

inline void check(int rc, const char* what) {
    TORCH_CHECK(rc == 0, what, " failed: ", rc == STP3_EINVAL ? "STP3_EINVAL" : rc == STP3_EUNSUP ? "STP3_EUNSUP"
                                          : rc == STP3_ENOSPACE ? "STP3_ENOSPACE" : "hipError ", rc);
}
// STP3_HOST_DRYRUN (scripts/host_overhead.py only): measure the host cost of this launch path on a machine
// without a GPU against a library whose entry points do nothing -- CPU tensors accepted, null stream.
const bool g_dryrun = getenv("STP3_HOST_DRYRUN") != nullptr;
inline void* stream() { return g_dryrun ? nullptr : (void*)c10::hip::getCurrentHIPStream().stream(); }
inline void need_gpu(const Tensor& t) {
    TORCH_CHECK(g_ready, "stp3_host.init(path to libstp3hip.so) has not been called");
    TORCH_CHECK(t.is_cuda() || g_dryrun,
                "stp3_amd operators run on the GPU only (got a CPU tensor); there is no fallback");
}
inline const void* ptr(const Tensor& t) { return t.defined() ? t.data_ptr() : nullptr; }
inline const float* fptr(const Tensor& t) { return t.defined() ? t.data_ptr<float>() : nullptr; }
inline Tensor f32(const Tensor& t) {
    if (!t.defined() || (t.scalar_type() == at::kFloat && t.is_contiguous())) return t;
    return t.detach().to(at::kFloat).contiguous();
}
inline Tensor opt(const c10::optional<Tensor>& t) { return t.has_value() ? *t : Tensor(); }

// (N, C, H, W) tensor -> tensor whose memory is [N][H*W][ld] channels-last (ops._rows_view)
Tensor rows_view(const Tensor& t, int64_t* ld) {
    const int64_t n = t.size(0), c = t.size(1), h = t.size(2), w = t.size(3);
    if (t.is_contiguous(at::MemoryFormat::ChannelsLast)) {
        *ld = c;
        return t;
    }
    const int64_t sn = t.stride(0), sc = t.stride(1), sh = t.stride(2), sw = t.stride(3);
    bool ok = !(c > 1 && sc != 1);
    int64_t l = c;
    if (ok) {
        l = w > 1 ? sw : (h > 1 ? sh : (n > 1 ? sn : c));
        ok = l >= c && (w == 1 || sw == l) && (h == 1 || sh == w * l) && (n == 1 || sn == h * w * l);
    }
    if (!ok) {
        *ld = c;
        return t.contiguous(at::MemoryFormat::ChannelsLast);
    }
    *ld = l;
    return t;
}

Tensor empty_cl(int64_t n, int64_t c, int64_t h, int64_t w, const at::TensorOptions& o) {
    return at::empty({n, c, h, w}, o.memory_format(at::MemoryFormat::ChannelsLast));
}

std::unordered_map<int, Tensor> g_bn_ws, g_conv_ws;
Tensor workspace(std::unordered_map<int, Tensor>& cache, const Tensor& like, int64_t need, int64_t floor_bytes) {
    const int dev = like.get_device();
    auto it = cache.find(dev);
    if (it == cache.end() || it->second.numel() < need) {
        Tensor ws = at::empty({std::max(need, floor_bytes)}, like.options().dtype(at::kByte));
        cache[dev] = ws;
        return ws;
    }
    return it->second;
}

int dtype_code(const Tensor& x) {
    if (x.scalar_type() == at::kBFloat16) return STP3_DTYPE_BF16;
    TORCH_CHECK(x.scalar_type() == at::kFloat, "stp3_host: float32 / bfloat16 only");
    return STP3_DTYPE_F32;
}

// =================================================================================================================
// fused BatchNorm + activation (+ per-sample bias, drop-connect scale, residual)      (ops._BnAct, world == 1)
// =================================================================================================================
struct BnActFn : public torch::autograd::Function<BnActFn> {
    using OptT = c10::optional<Tensor>;   // undefined tensors cannot go through Function::apply, optionals can
    static Tensor forward(AutogradContext* ctx, Tensor x, OptT weight_, OptT bias_, OptT res_, OptT sbias_, OptT oscale_,
                          OptT running_mean_, OptT running_var_, bool training, double momentum, double eps, int64_t act,
                          int64_t res_mode) {
        Tensor weight = opt(weight_), bias = opt(bias_), res = opt(res_), sbias = opt(sbias_), oscale = opt(oscale_);
        Tensor running_mean = opt(running_mean_), running_var = opt(running_var_);
        {   // autograd numbers its edges over the tensor arguments that are PRESENT, not over argument positions
            int64_t e = 1;
            std::vector<int64_t> edge{0, weight.defined() ? e++ : -1, bias.defined() ? e++ : -1, res.defined() ? e++ : -1,
                                      sbias.defined() ? e++ : -1};
            ctx->saved_data["edge"] = edge;
        }
        need_gpu(x);
        const int dt = dtype_code(x);
        const int64_t n = x.size(0), c = x.size(1), h = x.size(2), w = x.size(3);
        int64_t ldx, ldr = c;
        x = rows_view(x, &ldx);
        if (res.defined()) {
            TORCH_CHECK(res.sizes() == x.sizes(), "bn_act: residual shape mismatch");
            res = rows_view(res.scalar_type() == x.scalar_type() ? res : res.to(x.scalar_type()), &ldr);
        } else {
            res_mode = STP3_RES_NONE;
        }
        Tensor y = empty_cl(n, c, h, w, x.options());
        stp3_bn_dims d{(int32_t)n, (int32_t)(h * w), (int32_t)c, (int32_t)ldx, (int32_t)c, (int32_t)ldr, dt, (int32_t)act,
                       (int32_t)res_mode, sbias.defined() ? 1 : 0, oscale.defined() ? 1 : 0};
        Tensor gamma = f32(weight), beta = f32(bias), sb = f32(sbias), osc = f32(oscale);
        Tensor stat;
        if (training) {
            const int64_t need = n * STP3_BN_MAX_ROW_BLOCKS * 3 * c * 4;
            Tensor ws = workspace(g_bn_ws, x, need, 8 << 20);
            stat = at::empty({4 * c}, x.options().dtype(at::kFloat));
            check(api.stp3_bn_fwd_train(&d, x.data_ptr(), fptr(sb), ptr(res), fptr(osc), fptr(gamma), fptr(beta), (float)eps,
                                        (float)momentum, running_mean.defined() ? running_mean.data_ptr<float>() : nullptr,
                                        running_var.defined() ? running_var.data_ptr<float>() : nullptr,
                                        stat.data_ptr<float>(), ws.data_ptr(), (size_t)need, y.data_ptr(), stream()),
                  "stp3_bn_fwd_train");
        } else {
            Tensor rm = running_mean.detach().to(at::kFloat), rv = running_var.detach().to(at::kFloat);
            stat = at::cat({rm, rv, rm, at::rsqrt(rv + eps)});
            check(api.stp3_bn_apply_fwd(&d, x.data_ptr(), fptr(sb), ptr(res), fptr(osc), nullptr, 0.0, fptr(gamma), fptr(beta),
                                        (float)eps, 0.f, running_mean.data_ptr<float>(), running_var.data_ptr<float>(), nullptr,
                                        nullptr, y.data_ptr(), stream()),
                  "stp3_bn_apply_fwd");
        }
        ctx->save_for_backward({x, res_mode == STP3_RES_BEFORE_ACT ? res : Tensor(), sb, osc, gamma, beta, stat});
        ctx->saved_data["dims"] = std::vector<int64_t>{n, h * w, c, ldx, c, ldr, dt, act, res_mode, sbias.defined(), oscale.defined()};
        ctx->saved_data["training"] = training;
        ctx->saved_data["res_dtype"] = res.defined() ? (int64_t)res.scalar_type() : (int64_t)-1;
        ctx->saved_data["w_dtype"] = weight.defined() ? (int64_t)weight.scalar_type() : (int64_t)-1;
        ctx->saved_data["b_dtype"] = bias.defined() ? (int64_t)bias.scalar_type() : (int64_t)-1;
        return y;
    }

    static variable_list backward(AutogradContext* ctx, variable_list grads) {
        auto sv = ctx->get_saved_variables();
        Tensor x = sv[0], res = sv[1], sb = sv[2], osc = sv[3], gamma = sv[4], beta = sv[5], stat = sv[6];
        auto dv = ctx->saved_data["dims"].toIntVector();
        const bool training = ctx->saved_data["training"].toBool();
        const int64_t n = dv[0], rows = dv[1], c = dv[2];
        stp3_bn_dims d{(int32_t)dv[0], (int32_t)dv[1], (int32_t)dv[2], (int32_t)dv[3], (int32_t)dv[4], (int32_t)dv[5],
                       (int32_t)dv[6], (int32_t)dv[7], (int32_t)dv[8], (int32_t)dv[9], (int32_t)dv[10]};
        Tensor dy = grads[0];
        if (dy.scalar_type() != x.scalar_type()) dy = dy.to(x.scalar_type());
        int64_t ldy;
        dy = rows_view(dy, &ldy);
        if (ldy != c) dy = dy.contiguous(at::MemoryFormat::ChannelsLast);
        const int64_t need = n * STP3_BN_MAX_ROW_BLOCKS * 3 * c * 4;
        Tensor ws = workspace(g_bn_ws, x, need, 8 << 20);
        Tensor sumbuf = at::empty({(n + 1) * 3 * c}, x.options().dtype(at::kFloat));
        const int64_t sums_off = n * 3 * c;
        const float* mean_p = stat.data_ptr<float>() + 2 * c;
        const float* invstd_p = stat.data_ptr<float>() + 3 * c;
        Tensor dx = x.is_contiguous(at::MemoryFormat::ChannelsLast) ? at::empty_like(x)
                                                                    : at::empty_strided(x.sizes(), x.strides(), x.options());
        Tensor dres;
        auto edge = ctx->saved_data["edge"].toIntVector();
        auto wants = [&](int arg) { return edge[arg] >= 0 && ctx->needs_input_grad((size_t)edge[arg]); };
        const bool want_res = wants(3);
        if (d.res_mode == STP3_RES_BEFORE_ACT && want_res) dres = empty_cl(n, c, x.size(2), x.size(3), x.options());
        if (training && (!dres.defined() || d.ldr == c)) {
            check(api.stp3_bn_bwd_train(&d, dy.data_ptr(), x.data_ptr(), fptr(sb), ptr(res), fptr(osc), mean_p, invstd_p,
                                        fptr(gamma), fptr(beta), ws.data_ptr(), (size_t)need, sumbuf.data_ptr<float>(),
                                        dx.data_ptr(), dres.defined() ? dres.data_ptr() : nullptr, stream()),
                  "stp3_bn_bwd_train");
        } else {
            check(api.stp3_bn_bwd_reduce(&d, dy.data_ptr(), x.data_ptr(), fptr(sb), ptr(res), fptr(osc), mean_p, invstd_p,
                                         fptr(gamma), fptr(beta), ws.data_ptr(), (size_t)need, sumbuf.data_ptr<float>(),
                                         sumbuf.data_ptr<float>() + sums_off, stream()),
                  "stp3_bn_bwd_reduce");
            stp3_bn_dims bd = d;
            if (dres.defined() && d.ldr != c) {
                res = res.contiguous(at::MemoryFormat::ChannelsLast);
                bd.ldr = (int32_t)c;
            }
            check(api.stp3_bn_apply_bwd(&bd, dy.data_ptr(), x.data_ptr(), fptr(sb), ptr(res), fptr(osc), mean_p, invstd_p,
                                        fptr(gamma), fptr(beta), training ? sumbuf.data_ptr<float>() + sums_off : nullptr,
                                        training ? (double)std::max<int64_t>(n * rows, 1) : 1.0, dx.data_ptr(),
                                        dres.defined() ? dres.data_ptr() : nullptr, stream()),
                  "stp3_bn_apply_bwd");
        }
        Tensor sums = sumbuf.narrow(0, sums_off, 3 * c).view({3, c});
        Tensor dgamma, dbeta, dsbias;
        const int64_t wdt = ctx->saved_data["w_dtype"].toInt(), bdt = ctx->saved_data["b_dtype"].toInt();
        if (wdt >= 0 && wants(1)) dgamma = sums[1].to((at::ScalarType)wdt);
        if (bdt >= 0 && wants(2)) dbeta = sums[0].to((at::ScalarType)bdt);
        if (d.res_mode == STP3_RES_AFTER_ACT && want_res) dres = dy;
        const int64_t rdt = ctx->saved_data["res_dtype"].toInt();
        if (dres.defined() && rdt >= 0 && (int64_t)dres.scalar_type() != rdt) dres = dres.to((at::ScalarType)rdt);
        if (sb.defined() && wants(4)) {
            Tensor invstd = stat.narrow(0, 3 * c, c);
            Tensor sample = sumbuf.narrow(0, 0, sums_off).view({n, 3, c});
            Tensor g = gamma.defined() ? gamma * invstd : invstd;
            if (training) {
                Tensor k = sums / (double)(n * rows);
                dsbias = g * (sample.select(1, 0) - (double)rows * k[0] - sample.select(1, 2) * k[1]);
            } else {
                dsbias = g * sample.select(1, 0);
            }
        }
        return {dx, dgamma, dbeta, dres, dsbias, Tensor(), Tensor(), Tensor(), Tensor(), Tensor(), Tensor(), Tensor(), Tensor()};
    }
};

Tensor bn_act(Tensor x, c10::optional<Tensor> weight, c10::optional<Tensor> bias, c10::optional<Tensor> res,
              c10::optional<Tensor> sbias, c10::optional<Tensor> oscale, c10::optional<Tensor> running_mean,
              c10::optional<Tensor> running_var, bool training, double momentum, double eps, int64_t act, int64_t res_mode) {
    return BnActFn::apply(x, weight, bias, res, sbias, oscale, running_mean, running_var, training, momentum, eps, act, res_mode);
}

// =================================================================================================================
// dense convolution, bf16 MFMA implicit GEMM                                                 (ops._Conv2dMfma)
// =================================================================================================================
int64_t g_weight_epoch = 0;
int64_t g_wgrad_min_channels = 128;
struct WeightEntry {
    int64_t version = -1, epoch = -1;
    void* data = nullptr;
    Tensor wb, wt;
};
std::unordered_map<void*, WeightEntry> g_weights;

void invalidate_weight_cache() { ++g_weight_epoch; }
void set_wgrad_min_channels(int64_t v) { g_wgrad_min_channels = v; }

Tensor flipped(const Tensor& wb) { return wb.flip({2, 3}).transpose(0, 1).contiguous(at::MemoryFormat::ChannelsLast); }

// bf16 copies of a convolution weight (forward layout; tap-flipped / channel-swapped layout of the data gradient)
WeightEntry bf16_weights(const Tensor& weight, bool need_flipped, bool cacheable) {
    void* key = weight.unsafeGetTensorImpl();
    const int64_t ver = (int64_t)weight._version();
    if (cacheable) {
        auto it = g_weights.find(key);
        if (it != g_weights.end() && it->second.version == ver && it->second.epoch == g_weight_epoch &&
            it->second.data == weight.data_ptr()) {
            if (need_flipped && !it->second.wt.defined()) it->second.wt = flipped(it->second.wb);
            return it->second;
        }
    }
    WeightEntry e;
    e.version = ver; e.epoch = g_weight_epoch; e.data = weight.data_ptr();
    e.wb = weight.detach().to(at::kBFloat16).contiguous(at::MemoryFormat::ChannelsLast);
    if (need_flipped) e.wt = flipped(e.wb);
    if (cacheable) g_weights[key] = e;
    return e;
}

inline int64_t conv_out(int64_t size, int64_t k, int64_t stride, int64_t pad, int64_t dil) {
    return (size + 2 * pad - dil * (k - 1) - 1) / stride + 1;
}

Tensor conv_launch(Tensor x, const Tensor& wb, const Tensor& bias, int64_t stride, int64_t ph, int64_t pw, int64_t dh, int64_t dw,
                   at::ScalarType out_dtype) {
    const int64_t n = x.size(0), cin = x.size(1), h = x.size(2), w = x.size(3);
    const int64_t cout = wb.size(0), kh = wb.size(2), kw = wb.size(3);
    int64_t ldx;
    x = rows_view(x, &ldx);
    const int64_t ho = conv_out(h, kh, stride, ph, dh), wo = conv_out(w, kw, stride, pw, dw);
    Tensor y = empty_cl(n, cout, ho, wo, x.options().dtype(out_dtype));
    stp3_conv_dims d{(int32_t)n, (int32_t)h, (int32_t)w, (int32_t)cin, (int32_t)ho, (int32_t)wo, (int32_t)cout, (int32_t)kh,
                     (int32_t)kw, (int32_t)stride, (int32_t)ph, (int32_t)pw, (int32_t)dh, (int32_t)dw, (int32_t)ldx, (int32_t)cout,
                     out_dtype == at::kFloat ? STP3_DTYPE_F32 : STP3_DTYPE_BF16, bias.defined() ? 1 : 0};
    check(api.stp3_conv2d_fwd(&d, x.data_ptr(), wb.data_ptr(), fptr(bias), y.data_ptr(), stream()), "stp3_conv2d_fwd");
    return y;
}

Tensor conv_wgrad(Tensor dy, Tensor x, int64_t cout, int64_t cin, int64_t kh, int64_t kw, int64_t stride, int64_t ph, int64_t pw,
                  int64_t dh, int64_t dw) {
    int64_t ldx, ldy;
    x = rows_view(x, &ldx);
    dy = rows_view(dy, &ldy);
    stp3_conv_dims d{(int32_t)x.size(0), (int32_t)x.size(2), (int32_t)x.size(3), (int32_t)cin, (int32_t)dy.size(2),
                     (int32_t)dy.size(3), (int32_t)cout, (int32_t)kh, (int32_t)kw, (int32_t)stride, (int32_t)ph, (int32_t)pw,
                     (int32_t)dh, (int32_t)dw, (int32_t)ldx, (int32_t)ldy, STP3_DTYPE_F32, 0};
    size_t nbytes = 0;
    check(api.stp3_conv2d_wgrad_workspace(&d, &nbytes), "stp3_conv2d_wgrad_workspace");
    Tensor ws = workspace(g_conv_ws, x, (int64_t)nbytes, 64 << 20);
    Tensor dwt = empty_cl(cout, cin, kh, kw, x.options().dtype(at::kFloat));
    check(api.stp3_conv2d_wgrad(&d, dy.data_ptr(), x.data_ptr(), dwt.data_ptr<float>(), ws.data_ptr(), nbytes, stream()),
          "stp3_conv2d_wgrad");
    return dwt;
}

struct Conv2dFn : public torch::autograd::Function<Conv2dFn> {
    static Tensor forward(AutogradContext* ctx, Tensor x, Tensor weight, c10::optional<Tensor> bias_, int64_t stride, int64_t ph,
                          int64_t pw, int64_t dh, int64_t dw, bool out_f32) {
        Tensor bias = opt(bias_);
        need_gpu(x);
        if (x.scalar_type() != at::kBFloat16) x = x.to(at::kBFloat16);
        const bool cacheable = weight.is_leaf() && weight.requires_grad();
        WeightEntry e = bf16_weights(weight, false, cacheable);
        Tensor fb = f32(bias);
        Tensor y = conv_launch(x, e.wb, fb, stride, ph, pw, dh, dw, out_f32 ? at::kFloat : at::kBFloat16);
        ctx->save_for_backward({x, e.wb, cacheable ? weight : Tensor()});
        ctx->saved_data["cfg"] = std::vector<int64_t>{stride, ph, pw, dh, dw, bias.defined(), (int64_t)weight.scalar_type(),
                                                      bias.defined() ? (int64_t)bias.scalar_type() : -1};
        return y;
    }

    static variable_list backward(AutogradContext* ctx, variable_list grads) {
        auto sv = ctx->get_saved_variables();
        Tensor x = sv[0], wb = sv[1], weight = sv[2];
        auto cfg = ctx->saved_data["cfg"].toIntVector();
        const int64_t stride = cfg[0], ph = cfg[1], pw = cfg[2], dh = cfg[3], dw = cfg[4];
        const bool has_bias = cfg[5] != 0;
        const auto wdt = (at::ScalarType)cfg[6];
        const int64_t cout = wb.size(0), cin = wb.size(1), kh = wb.size(2), kw = wb.size(3);
        Tensor dy = grads[0].to(at::kBFloat16).contiguous(at::MemoryFormat::ChannelsLast);
        Tensor dx, dwt, db;
        const int64_t bph = dh * (kh - 1) - ph, bpw = dw * (kw - 1) - pw;
        const bool need_dx = ctx->needs_input_grad(0);
        const bool hip_dx = need_dx && cout % 8 == 0 && bph >= 0 && bpw >= 0;
        if (hip_dx) {
            Tensor wt = weight.defined() ? bf16_weights(weight, true, true).wt : flipped(wb);
            Tensor g = dy;
            if (stride > 1) {
                const int64_t uh = x.size(2) + 2 * ph - dh * (kh - 1), uw = x.size(3) + 2 * pw - dw * (kw - 1);
                g = empty_cl(x.size(0), cout, uh, uw, dy.options()).zero_();
                using at::indexing::Slice;
                g.index({Slice(), Slice(), Slice(0, c10::nullopt, stride), Slice(0, c10::nullopt, stride)})
                    .index({Slice(), Slice(), Slice(0, dy.size(2)), Slice(0, dy.size(3))})
                    .copy_(dy);
            }
            dx = conv_launch(g, wt, Tensor(), 1, bph, bpw, dh, dw, at::kBFloat16);
        }
        const bool need_dw = ctx->needs_input_grad(1), need_db = has_bias && ctx->needs_input_grad(2);
        const bool hip_dw = need_dw && cin % 4 == 0 && cout % 4 == 0 && g_wgrad_min_channels <= std::min(cin, cout);
        if (hip_dw) {
            dwt = conv_wgrad(dy, x, cout, cin, kh, kw, stride, ph, pw, dh, dw).to(wdt);
            if (need_db) db = dy.to(at::kFloat).sum({0, 2, 3}).to((at::ScalarType)cfg[7]);
        }
        std::array<bool, 3> mask{need_dx && !hip_dx, need_dw && !hip_dw, need_db && !hip_dw};
        if (mask[0] || mask[1] || mask[2]) {
            Tensor xd = x.is_contiguous(at::MemoryFormat::ChannelsLast) ? x : x.contiguous(at::MemoryFormat::ChannelsLast);
            c10::optional<at::IntArrayRef> bias_sizes;
            std::vector<int64_t> bs{cout};
            if (has_bias) bias_sizes = at::IntArrayRef(bs);
            auto r = at::convolution_backward(dy, xd, wb, bias_sizes, {stride, stride}, {ph, pw}, {dh, dw}, false, {0, 0}, 1, mask);
            if (mask[0]) dx = std::get<0>(r);
            if (mask[1]) dwt = std::get<1>(r).to(wdt);
            if (mask[2]) db = std::get<2>(r).to((at::ScalarType)cfg[7]);
        }
        return {dx, dwt, db, Tensor(), Tensor(), Tensor(), Tensor(), Tensor(), Tensor()};
    }
};

Tensor conv2d(Tensor x, Tensor weight, c10::optional<Tensor> bias, int64_t stride, int64_t ph, int64_t pw, int64_t dh, int64_t dw,
              bool out_f32) {
    return Conv2dFn::apply(x, weight, bias, stride, ph, pw, dh, dw, out_f32);
}

// =================================================================================================================
// depthwise convolution                                                                    (ops._DepthwiseConv2d)
// =================================================================================================================
struct DwConvFn : public torch::autograd::Function<DwConvFn> {
    static Tensor forward(AutogradContext* ctx, Tensor x, Tensor weight, int64_t stride, int64_t left, int64_t right, int64_t top,
                          int64_t bottom) {
        need_gpu(x);
        const int64_t c = weight.size(0), k = weight.size(2);
        const int64_t n = x.size(0), h = x.size(2), w = x.size(3);
        const int64_t ho = (h + top + bottom - k) / stride + 1, wo = (w + left + right - k) / stride + 1;
        x = x.contiguous(at::MemoryFormat::ChannelsLast);
        Tensor wt = weight.detach().to(at::kFloat).reshape({c, k * k}).t().contiguous();   // [K*K][C]
        Tensor y = empty_cl(n, c, ho, wo, x.options());
        stp3_dwconv_dims d{(int32_t)n, (int32_t)h, (int32_t)w, (int32_t)c, (int32_t)ho, (int32_t)wo, (int32_t)k, (int32_t)stride,
                           (int32_t)top, (int32_t)left, dtype_code(x)};
        check(api.stp3_dwconv2d_fwd(&d, x.data_ptr(), wt.data_ptr<float>(), y.data_ptr(), stream()), "stp3_dwconv2d_fwd");
        ctx->save_for_backward({x, wt});
        ctx->saved_data["d"] = std::vector<int64_t>{n, h, w, c, ho, wo, k, stride, top, left, d.dtype, (int64_t)weight.scalar_type()};
        return y;
    }

    static variable_list backward(AutogradContext* ctx, variable_list grads) {
        auto sv = ctx->get_saved_variables();
        Tensor x = sv[0], wt = sv[1];
        auto v = ctx->saved_data["d"].toIntVector();
        stp3_dwconv_dims d{(int32_t)v[0], (int32_t)v[1], (int32_t)v[2], (int32_t)v[3], (int32_t)v[4], (int32_t)v[5], (int32_t)v[6],
                           (int32_t)v[7], (int32_t)v[8], (int32_t)v[9], (int32_t)v[10]};
        Tensor dy = grads[0].contiguous(at::MemoryFormat::ChannelsLast);
        if (dy.scalar_type() != x.scalar_type()) dy = dy.to(x.scalar_type());
        Tensor dx, dw;
        if (ctx->needs_input_grad(0)) {
            dx = at::empty_like(x, x.options().memory_format(at::MemoryFormat::ChannelsLast));
            check(api.stp3_dwconv2d_bwd_data(&d, dy.data_ptr(), wt.data_ptr<float>(), dx.data_ptr(), stream()),
                  "stp3_dwconv2d_bwd_data");
        }
        if (ctx->needs_input_grad(1)) {
            size_t nbytes = 0;
            check(api.stp3_dwconv2d_bwd_weight_workspace(&d, &nbytes), "stp3_dwconv2d_bwd_weight_workspace");
            Tensor ws = at::empty({(int64_t)nbytes}, x.options().dtype(at::kByte));
            Tensor dwt = at::empty_like(wt);
            check(api.stp3_dwconv2d_bwd_weight(&d, x.data_ptr(), dy.data_ptr(), dwt.data_ptr<float>(), ws.data_ptr(), nbytes, stream()),
                  "stp3_dwconv2d_bwd_weight");
            dw = dwt.t().reshape({v[3], 1, v[6], v[6]}).to((at::ScalarType)v[11]);
        }
        return {dx, dw, Tensor(), Tensor(), Tensor(), Tensor(), Tensor()};
    }
};

Tensor depthwise_conv2d(Tensor x, Tensor weight, int64_t stride, int64_t left, int64_t right, int64_t top, int64_t bottom) {
    return DwConvFn::apply(x, weight, stride, left, right, top, bottom);
}

}  // namespace

PYBIND11_MODULE(TORCH_EXTENSION_NAME, m) {
    m.doc() = "C++ launch path of the stp3_amd custom operators (experimental)";
    m.def("init", &init, "dlopen libstp3hip.so and resolve the C ABI");
    m.def("bn_act", &bn_act);
    m.def("conv2d", &conv2d);
    m.def("depthwise_conv2d", &depthwise_conv2d);
    m.def("invalidate_weight_cache", &invalidate_weight_cache);
    m.def("set_wgrad_min_channels", &set_wgrad_min_channels);
}
