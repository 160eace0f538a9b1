// PyTorch dispatcher registration of the hot-path kernels: TORCH_LIBRARY(gamer, ...) wrappers over the C ABI of
// libgamer_hip.so (include/gamer_hip.h), as SURVEY.md section 8(b) specifies for the drop-in boundary
// ("custom ops take/return at::Tensor; launch on the current HIP stream; TORCH_CHECK -> RuntimeError").
//
// The ops are thin: argument checks, output allocation through the caching allocator, one C-ABI call on
// c10::hip::getCurrentHIPStream().  fp32 and bf16 activations dispatch on the tensor dtype to the gamer_*_bf16 twin.
// They make the kernels visible to the dispatcher (profiler op names, torch.compile as opaque custom ops, torch.library
// tooling); the engine's own step drives the same entry points through ctypes without the dispatcher in between.
//
// Built by gamer_amd/build.py with the host compiler against the installed torch headers into
// gamer_amd/lib/libgamer_torch.so; loaded by gamer_amd/torch_ops.py (torch.ops.load_library).
#include <ATen/ATen.h>
#include <c10/hip/HIPStream.h>
#include <torch/library.h>

#include <tuple>

#include "../../include/gamer_hip.h"

namespace {

using at::Tensor;
using c10::optional;

void* cur_stream() { return (void*)c10::hip::getCurrentHIPStream().stream(); }

void check(int rc, const char* what) { TORCH_CHECK(rc == 0, what, " failed (rc=", rc, "): ", gamer_last_error()); }

bool is_bf16(const Tensor& t) { return t.scalar_type() == at::kBFloat16; }

void need(const Tensor& t, const char* name, at::ScalarType st) {
    TORCH_CHECK(t.is_cuda(), name, " must be on the HIP device, got ", t.device());
    TORCH_CHECK(t.scalar_type() == st, name, " must be ", st, ", got ", t.scalar_type());
    TORCH_CHECK(t.is_contiguous(), name, " must be contiguous, got strides ", t.strides());
}
void need_act(const Tensor& t, const char* name) {
    TORCH_CHECK(t.is_cuda() && t.is_contiguous(), name, " must be a contiguous HIP tensor");
    TORCH_CHECK(t.scalar_type() == at::kFloat || t.scalar_type() == at::kBFloat16, name, " must be float32 or bfloat16, got ",
                t.scalar_type());
}
template <typename T> T* ptr(const Tensor& t) { return reinterpret_cast<T*>(t.data_ptr()); }
template <typename T> T* optr(const optional<Tensor>& t) { return t.has_value() ? reinterpret_cast<T*>(t->data_ptr()) : nullptr; }

// ---- RMSNorm (Qwen3RMSNorm; model.py:205,222,239,869) ------------------------------------------------------------
Tensor rmsnorm_fwd(const Tensor& x, const Tensor& w, double eps, bool bf16_out) {
    need(x, "x", at::kFloat); need(w, "w", at::kFloat);
    TORCH_CHECK(x.dim() == 2 && w.dim() == 1 && w.size(0) == x.size(1), "rmsnorm_fwd: x [T,H], w [H]; got ", x.sizes(), " ", w.sizes());
    const int T = (int)x.size(0), H = (int)x.size(1);
    Tensor y = at::empty({T, H}, x.options().dtype(bf16_out ? at::kBFloat16 : at::kFloat));
    if (bf16_out) check(gamer_rmsnorm_fwd_bf16(ptr<float>(x), ptr<float>(w), T, H, (float)eps, nullptr, ptr<gamer_bf16>(y), H, cur_stream()), "gamer_rmsnorm_fwd_bf16");
    else check(gamer_rmsnorm_fwd(ptr<float>(x), ptr<float>(w), T, H, (float)eps, nullptr, ptr<float>(y), H, cur_stream()), "gamer_rmsnorm_fwd");
    return y;
}

std::tuple<Tensor, Tensor> rmsnorm_bwd(const Tensor& x, const Tensor& w, const Tensor& dy, double eps) {
    need(x, "x", at::kFloat); need(w, "w", at::kFloat); need_act(dy, "dy");
    TORCH_CHECK(x.dim() == 2 && dy.sizes() == x.sizes(), "rmsnorm_bwd: x, dy [T,H]");
    const int T = (int)x.size(0), H = (int)x.size(1), NP = 512;
    Tensor dx = at::empty_like(x), part = at::empty({NP, H}, x.options()), dw = at::zeros({H}, x.options());
    if (is_bf16(dy)) check(gamer_rmsnorm_bwd_bf16(ptr<float>(x), ptr<float>(w), ptr<gamer_bf16>(dy), H, nullptr, T, H, (float)eps, 0, ptr<float>(dx), ptr<float>(part), NP, nullptr, nullptr, 0.f, 0, cur_stream()), "gamer_rmsnorm_bwd_bf16");
    else check(gamer_rmsnorm_bwd(ptr<float>(x), ptr<float>(w), ptr<float>(dy), H, nullptr, T, H, (float)eps, 0, ptr<float>(dx), ptr<float>(part), NP, nullptr, nullptr, 0.f, 0, cur_stream()), "gamer_rmsnorm_bwd");
    check(gamer_colsum_reduce(ptr<float>(part), NP, H, 0, ptr<float>(dw), cur_stream()), "gamer_colsum_reduce");
    return {dx, dw};
}

// ---- nn.Linear forward: y = x W^T (model.py:93-99,145-149,1001; FFN.py:25-27) --------------------------------------
Tensor linear(const Tensor& x, const Tensor& w) {
    need_act(x, "x"); need_act(w, "w");
    TORCH_CHECK(x.dim() == 2 && w.dim() == 2 && x.size(1) == w.size(1) && x.scalar_type() == w.scalar_type(),
                "linear: x [M,K], w [N,K] of one dtype; got ", x.sizes(), " ", w.sizes());
    const int M = (int)x.size(0), K = (int)x.size(1), N = (int)w.size(0);
    Tensor y = at::empty({M, N}, x.options());
    if (is_bf16(x)) {
        gamer_gemm_bf16_desc d{};
        d.A = ptr<gamer_bf16>(x); d.lda = K; d.B = ptr<gamer_bf16>(w); d.ldb = K; d.C = y.data_ptr(); d.ldc = N;
        d.M = M; d.N = N; d.K = K; d.groups = 1;
        check(gamer_gemm_bf16(&d, cur_stream()), "gamer_gemm_bf16");
    } else {
        gamer_gemm_desc d{};
        d.A = ptr<float>(x); d.a_rs = K; d.a_ks = 1; d.B = ptr<float>(w); d.b_rs = K; d.b_ks = 1; d.C = ptr<float>(y); d.ldc = N;
        d.M = M; d.N = N; d.K = K; d.alpha = 1.f; d.groups = 1;
        check(gamer_gemm_f32(&d, cur_stream()), "gamer_gemm_f32");
    }
    return y;
}

// ---- q/k per-head RMSNorm + RoPE (+ behaviour biases) (model.py:88-101) -------------------------------------------
std::tuple<Tensor, Tensor> qkv_rope_fwd(Tensor qkv, int64_t S, int64_t nq, int64_t nkv, const Tensor& wq, const Tensor& wk,
                                        double eps, const Tensor& cos_t, const Tensor& sin_t, const optional<Tensor>& bias_q,
                                        const optional<Tensor>& bias_k, const optional<Tensor>& bias_v,
                                        const optional<Tensor>& act_idx) {
    need_act(qkv, "qkv"); need(wq, "wq", at::kFloat); need(wk, "wk", at::kFloat); need(cos_t, "cos", at::kFloat); need(sin_t, "sin", at::kFloat);
    TORCH_CHECK(qkv.dim() == 2 && qkv.size(1) == (nq + 2 * nkv) * 64, "qkv_rope_fwd: qkv [T,(nq+2nkv)*64], got ", qkv.sizes());
    if (act_idx.has_value()) need(*act_idx, "act_idx", at::kInt);
    const int T = (int)qkv.size(0);
    Tensor q = at::empty({T, nq * 64}, qkv.options()), k = at::empty({T, nkv * 64}, qkv.options());
    if (is_bf16(qkv))
        check(gamer_qknorm_rope_fwd_bf16(ptr<gamer_bf16>(qkv), T, (int)S, (int)nq, (int)nkv, ptr<float>(wq), ptr<float>(wk), (float)eps, ptr<float>(cos_t), ptr<float>(sin_t), optr<float>(bias_q), optr<float>(bias_k), optr<float>(bias_v), optr<int32_t>(act_idx), ptr<gamer_bf16>(q), ptr<gamer_bf16>(k), nullptr, cur_stream()), "gamer_qknorm_rope_fwd_bf16");
    else
        check(gamer_qknorm_rope_fwd(ptr<float>(qkv), T, (int)S, (int)nq, (int)nkv, ptr<float>(wq), ptr<float>(wk), (float)eps, ptr<float>(cos_t), ptr<float>(sin_t), optr<float>(bias_q), optr<float>(bias_k), optr<float>(bias_v), optr<int32_t>(act_idx), ptr<float>(q), ptr<float>(k), nullptr, cur_stream()), "gamer_qknorm_rope_fwd");
    return {q, k};
}

// ---- multi-behaviour attention (model.py:133-143 + the two mask builders) -------------------------------------------
// kl / ql / row_empty / tile_empty come from gamer_router_fwd (ql undefined = self attention).
std::tuple<Tensor, Tensor> mb_attention_fwd(const Tensor& q, const Tensor& k, const Tensor& v, const Tensor& kl,
                                            const optional<Tensor>& ql, const Tensor& row_empty, const Tensor& tile_empty,
                                            int64_t B, int64_t S, int64_t nq, int64_t nkv, double scale, double p_drop, int64_t seed) {
    need_act(q, "q"); need_act(k, "k"); need(kl, "kl", at::kInt); need(row_empty, "row_empty", at::kInt); need(tile_empty, "tile_empty", at::kInt);
    TORCH_CHECK(v.is_cuda() && v.stride(1) == 1 && v.scalar_type() == q.scalar_type(), "v must be a HIP tensor with unit column stride (it may be a column slice of qkv)");
    const int T = (int)(B * S);
    TORCH_CHECK(q.size(0) == T && k.size(0) == T && v.size(0) == T, "mb_attention_fwd: q/k/v need B*S rows");
    Tensor o = at::empty({T, nq * 64}, q.options()), lse = at::empty({B, nq, S}, q.options().dtype(at::kFloat));
    if (is_bf16(q))
        check(gamer_attn_fwd_bf16(ptr<gamer_bf16>(q), (int)q.stride(0), ptr<gamer_bf16>(k), (int)k.stride(0), ptr<gamer_bf16>(v), (int)v.stride(0), ptr<int32_t>(kl), optr<int32_t>(ql), (int)B, (int)S, (int)nq, (int)nkv, (float)scale, (float)p_drop, (uint64_t)seed, ptr<gamer_bf16>(o), ptr<float>(lse), nullptr, nullptr, nullptr, nullptr, cur_stream()), "gamer_attn_fwd_bf16");
    else
        check(gamer_attn_fwd(ptr<float>(q), (int)q.stride(0), ptr<float>(k), (int)k.stride(0), ptr<float>(v), (int)v.stride(0), ptr<int32_t>(kl), optr<int32_t>(ql), ptr<int32_t>(row_empty), ptr<int32_t>(tile_empty), (int)B, (int)S, (int)nq, (int)nkv, (float)scale, (float)p_drop, (uint64_t)seed, ptr<float>(o), ptr<float>(lse), nullptr, nullptr, nullptr, 0, nullptr, cur_stream()), "gamer_attn_fwd");
    return {o, lse};
}

std::tuple<Tensor, Tensor, Tensor> mb_attention_bwd(const Tensor& q, const Tensor& k, const Tensor& v, const Tensor& o,
                                                    const Tensor& d_o, const Tensor& lse, const Tensor& kl,
                                                    const optional<Tensor>& ql, const Tensor& row_empty,
                                                    const Tensor& tile_empty, int64_t B, int64_t S, int64_t nq, int64_t nkv,
                                                    double scale, double p_drop, int64_t seed) {
    need_act(q, "q"); need_act(k, "k"); need_act(o, "o"); need_act(d_o, "d_o"); need(lse, "lse", at::kFloat);
    const int T = (int)(B * S);
    Tensor dq = at::empty({T, nq * 64}, q.options()), dk = at::empty({T, nkv * 64}, q.options()), dv = at::empty({T, nkv * 64}, q.options());
    Tensor delta = at::empty({B, nq, S}, lse.options());
    if (is_bf16(q))
        check(gamer_attn_bwd_bf16(ptr<gamer_bf16>(q), (int)q.stride(0), ptr<gamer_bf16>(k), (int)k.stride(0), ptr<gamer_bf16>(v), (int)v.stride(0), ptr<gamer_bf16>(o), ptr<gamer_bf16>(d_o), ptr<float>(lse), ptr<int32_t>(kl), optr<int32_t>(ql), (int)B, (int)S, (int)nq, (int)nkv, (float)scale, (float)p_drop, (uint64_t)seed, ptr<float>(delta), ptr<gamer_bf16>(dq), (int)nq * 64, ptr<gamer_bf16>(dk), (int)nkv * 64, ptr<gamer_bf16>(dv), (int)nkv * 64, nullptr, 0, nullptr, nullptr, nullptr, cur_stream()), "gamer_attn_bwd_bf16");
    else
        check(gamer_attn_bwd(ptr<float>(q), (int)q.stride(0), ptr<float>(k), (int)k.stride(0), ptr<float>(v), (int)v.stride(0), ptr<float>(o), ptr<float>(d_o), ptr<float>(lse), ptr<int32_t>(kl), optr<int32_t>(ql), ptr<int32_t>(row_empty), ptr<int32_t>(tile_empty), (int)B, (int)S, (int)nq, (int)nkv, (float)scale, (float)p_drop, (uint64_t)seed, ptr<float>(delta), ptr<float>(dq), (int)nq * 64, ptr<float>(dk), (int)nkv * 64, ptr<float>(dv), (int)nkv * 64, nullptr, nullptr, nullptr, nullptr, nullptr, 0, cur_stream()), "gamer_attn_bwd");
    return {dq, dk, dv};
}

// ---- SwiGLU of the position-routed experts (FFN.py:25-27) ----------------------------------------------------------
Tensor swiglu_fwd(const Tensor& g, const Tensor& u, double p_drop, int64_t seed) {
    need_act(g, "g"); need_act(u, "u");
    TORCH_CHECK(g.sizes() == u.sizes() && g.scalar_type() == u.scalar_type() && g.numel() % 4 == 0, "swiglu_fwd: g, u of one shape, numel % 4 == 0");
    Tensor hm = at::empty_like(g);
    if (is_bf16(g)) check(gamer_swiglu_fwd_bf16(ptr<gamer_bf16>(g), ptr<gamer_bf16>(u), g.numel(), (float)p_drop, (uint64_t)seed, ptr<gamer_bf16>(hm), cur_stream()), "gamer_swiglu_fwd_bf16");
    else check(gamer_swiglu_fwd(ptr<float>(g), ptr<float>(u), g.numel(), (float)p_drop, (uint64_t)seed, ptr<float>(hm), cur_stream()), "gamer_swiglu_fwd");
    return hm;
}

std::tuple<Tensor, Tensor> swiglu_bwd(const Tensor& g, const Tensor& u, const Tensor& dhm, double p_drop, int64_t seed) {
    need_act(g, "g"); need_act(u, "u"); need_act(dhm, "dhm");
    Tensor dg = g.clone(), du = u.clone();                       // the kernel works in place
    if (is_bf16(g)) check(gamer_swiglu_bwd_bf16(ptr<gamer_bf16>(dg), ptr<gamer_bf16>(du), ptr<gamer_bf16>(dhm), g.numel(), (float)p_drop, (uint64_t)seed, cur_stream()), "gamer_swiglu_bwd_bf16");
    else check(gamer_swiglu_bwd(ptr<float>(dg), ptr<float>(du), ptr<float>(dhm), g.numel(), (float)p_drop, (uint64_t)seed, cur_stream()), "gamer_swiglu_bwd");
    return {dg, du};
}

// ---- position-routed SwiGLU experts as one op (FFN.py:53-72 in expert-sorted order) ---------------------------------------
// hin [T, din] fp32 rows sorted by expert (gamer_expert_lists), w_gate / w_up [E*I, din] stacked per expert, offsets int32 [E+1]:
// g = hin W_gate_e^T, u = hin W_up_e^T as grouped GEMMs over the expert segments, hm = dropout(silu(g) * u).
void grouped_gemm_f32(const float* A, int64_t a_rs, int64_t a_ks, const float* B, int64_t b_rs, int64_t b_ks, float* C, int64_t ldc,
                      int M, int N, int K, int groups, int mode, const int32_t* offs, int64_t strideB, int64_t strideC, int kchunk,
                      int accumulate, const char* what) {
    gamer_gemm_desc d{};
    d.A = A; d.a_rs = a_rs; d.a_ks = a_ks; d.B = B; d.b_rs = b_rs; d.b_ks = b_ks; d.C = C; d.ldc = ldc;
    d.M = M; d.N = N; d.K = K; d.alpha = 1.f; d.accumulate = accumulate; d.groups = groups; d.group_mode = mode;
    d.group_offsets = offs; d.strideB = strideB; d.strideC = strideC; d.kchunk = kchunk;
    check(gamer_gemm_f32(&d, cur_stream()), what);
}
std::tuple<Tensor, Tensor, Tensor> routed_swiglu_fwd(const Tensor& hin, const Tensor& w_gate, const Tensor& w_up, const Tensor& offsets,
                                                     double p_drop, int64_t seed) {
    need(hin, "hin", at::kFloat); need(w_gate, "w_gate", at::kFloat); need(w_up, "w_up", at::kFloat); need(offsets, "offsets", at::kInt);
    TORCH_CHECK(hin.dim() == 2 && w_gate.dim() == 2 && w_gate.sizes() == w_up.sizes() && w_gate.size(1) == hin.size(1) && offsets.dim() == 1 &&
                offsets.numel() >= 2 && w_gate.size(0) % (offsets.numel() - 1) == 0,
                "routed_swiglu_fwd: hin [T,din], w_gate / w_up [E*I,din], offsets [E+1]; got ", hin.sizes(), " ", w_gate.sizes(), " ", offsets.sizes());
    const int T = (int)hin.size(0), din = (int)hin.size(1), E = (int)offsets.numel() - 1, I = (int)(w_gate.size(0) / E);
    Tensor g = at::empty({T, I}, hin.options()), u = at::empty({T, I}, hin.options()), hm = at::empty({T, I}, hin.options());
    grouped_gemm_f32(ptr<float>(hin), din, 1, ptr<float>(w_gate), din, 1, ptr<float>(g), I, T, I, din, E, 0, ptr<int32_t>(offsets), (int64_t)I * din, 0, 0, 0, "gamer_gemm_f32 (gate)");
    grouped_gemm_f32(ptr<float>(hin), din, 1, ptr<float>(w_up), din, 1, ptr<float>(u), I, T, I, din, E, 0, ptr<int32_t>(offsets), (int64_t)I * din, 0, 0, 0, "gamer_gemm_f32 (up)");
    check(gamer_swiglu_fwd(ptr<float>(g), ptr<float>(u), g.numel(), (float)p_drop, (uint64_t)seed, ptr<float>(hm), cur_stream()), "gamer_swiglu_fwd");
    return {g, u, hm};
}
// (dhin, dw_gate, dw_up) from d(hm); g, u, hin as returned / passed by routed_swiglu_fwd
std::tuple<Tensor, Tensor, Tensor> routed_swiglu_bwd(const Tensor& g, const Tensor& u, const Tensor& dhm, const Tensor& hin,
                                                     const Tensor& w_gate, const Tensor& w_up, const Tensor& offsets, double p_drop,
                                                     int64_t seed) {
    need(g, "g", at::kFloat); need(u, "u", at::kFloat); need(dhm, "dhm", at::kFloat); need(hin, "hin", at::kFloat);
    need(w_gate, "w_gate", at::kFloat); need(w_up, "w_up", at::kFloat); need(offsets, "offsets", at::kInt);
    TORCH_CHECK(g.sizes() == u.sizes() && g.sizes() == dhm.sizes() && hin.size(0) == g.size(0) && w_gate.sizes() == w_up.sizes(),
                "routed_swiglu_bwd: shapes");
    const int T = (int)hin.size(0), din = (int)hin.size(1), E = (int)offsets.numel() - 1, I = (int)g.size(1);
    Tensor dg = g.clone(), du = u.clone();                       // gamer_swiglu_bwd works in place
    check(gamer_swiglu_bwd(ptr<float>(dg), ptr<float>(du), ptr<float>(dhm), g.numel(), (float)p_drop, (uint64_t)seed, cur_stream()), "gamer_swiglu_bwd");
    Tensor dwg = at::zeros_like(w_gate), dwu = at::zeros_like(w_up), dhin = at::empty_like(hin);
    const int kchunk = 1024;                                     // token chunk of the split-K weight gradient (gamer_amd/ops.py: pick_kchunk)
    grouped_gemm_f32(ptr<float>(dg), 1, I, ptr<float>(hin), 1, din, ptr<float>(dwg), din, I, din, T, E, 1, ptr<int32_t>(offsets), 0, (int64_t)I * din, kchunk, 1, "gamer_gemm_f32 (gate wgrad)");
    grouped_gemm_f32(ptr<float>(du), 1, I, ptr<float>(hin), 1, din, ptr<float>(dwu), din, I, din, T, E, 1, ptr<int32_t>(offsets), 0, (int64_t)I * din, kchunk, 1, "gamer_gemm_f32 (up wgrad)");
    grouped_gemm_f32(ptr<float>(dg), I, 1, ptr<float>(w_gate), 1, din, ptr<float>(dhin), din, T, din, I, E, 0, ptr<int32_t>(offsets), (int64_t)I * din, 0, 0, 0, "gamer_gemm_f32 (gate dgrad)");
    grouped_gemm_f32(ptr<float>(du), I, 1, ptr<float>(w_up), 1, din, ptr<float>(dhin), din, T, din, I, E, 0, ptr<int32_t>(offsets), (int64_t)I * din, 0, 0, 1, "gamer_gemm_f32 (up dgrad)");
    return {dhin, dwg, dwu};
}
// ---- tied head loss: temperature + shifted cross entropy (model.py:904-922) -----------------------------------------
// logits [B*S, ld] are divided by the temperature IN PLACE (as upstream); returns (sum CE, #targets, per-row lse)
std::tuple<Tensor, Tensor, Tensor> lmhead_ce_fwd(Tensor logits, const Tensor& labels, int64_t V, double temperature) {
    need_act(logits, "logits"); need(labels, "labels", at::kLong);
    TORCH_CHECK(labels.dim() == 2 && logits.dim() == 2 && logits.size(0) == labels.numel(), "lmhead_ce_fwd: logits [B*S, ld], labels [B,S]");
    const int B = (int)labels.size(0), S = (int)labels.size(1), T = B * S;
    auto f32 = logits.options().dtype(at::kFloat);
    Tensor lse = at::empty({T}, f32), row = at::empty({T}, f32), sum = at::zeros({1}, f32), cnt = at::zeros({1}, f32);
    if (is_bf16(logits)) check(gamer_ce_fwd_bf16(ptr<gamer_bf16>(logits), (int)logits.stride(0), ptr<int64_t>(labels), B, S, (int)V, (float)temperature, -100, ptr<float>(lse), ptr<float>(row), ptr<float>(sum), ptr<float>(cnt), cur_stream()), "gamer_ce_fwd_bf16");
    else check(gamer_ce_fwd(ptr<float>(logits), (int)logits.stride(0), ptr<int64_t>(labels), B, S, (int)V, (float)temperature, -100, ptr<float>(lse), ptr<float>(row), ptr<float>(sum), ptr<float>(cnt), cur_stream()), "gamer_ce_fwd");
    return {sum, cnt, lse};
}

// logits (already scaled by lmhead_ce_fwd) <- d(loss)/d(raw logits) for loss = sum CE / denom, times dloss
void lmhead_ce_bwd(Tensor logits, const Tensor& labels, const Tensor& lse, int64_t V, double temperature, const Tensor& denom,
                   const Tensor& dloss) {
    need_act(logits, "logits"); need(labels, "labels", at::kLong); need(lse, "lse", at::kFloat); need(denom, "denom", at::kFloat); need(dloss, "dloss", at::kFloat);
    const int B = (int)labels.size(0), S = (int)labels.size(1);
    if (is_bf16(logits)) check(gamer_ce_bwd_bf16(ptr<gamer_bf16>(logits), (int)logits.stride(0), ptr<int64_t>(labels), B, S, (int)V, (float)temperature, -100, ptr<float>(lse), ptr<float>(denom), 0.f, 1.f, ptr<float>(dloss), cur_stream()), "gamer_ce_bwd_bf16");
    else check(gamer_ce_bwd(ptr<float>(logits), (int)logits.stride(0), ptr<int64_t>(labels), B, S, (int)V, (float)temperature, -100, ptr<float>(lse), ptr<float>(denom), 0.f, 1.f, ptr<float>(dloss), cur_stream()), "gamer_ce_bwd");
}

// ---- HF Trainer update: clip_grad_norm_ + AdamW over flat fp32 buffers (train_SMB_decoder.py:396-428) ---------------
Tensor fused_adamw_clip(Tensor p, const Tensor& g, Tensor m, Tensor v, int64_t n_decay, double lr, double beta1, double beta2,
                        double eps, double weight_decay, int64_t step, double max_norm, double grad_scale) {
    need(p, "p", at::kFloat); need(g, "g", at::kFloat); need(m, "m", at::kFloat); need(v, "v", at::kFloat);
    TORCH_CHECK(p.dim() == 1 && g.sizes() == p.sizes() && m.sizes() == p.sizes() && v.sizes() == p.sizes(), "fused_adamw_clip: flat buffers of one length");
    const int NP = 1024;
    Tensor part = at::empty({NP}, p.options()), norm = at::zeros({1}, p.options());
    check(gamer_sumsq(ptr<float>(g), g.numel(), ptr<float>(part), NP, cur_stream()), "gamer_sumsq");
    check(gamer_adamw(ptr<float>(p), ptr<float>(g), ptr<float>(m), ptr<float>(v), p.numel(), n_decay, (float)lr, (float)beta1, (float)beta2, (float)eps, (float)weight_decay, (int)step, (float)max_norm, (float)grad_scale, ptr<float>(part), NP, ptr<float>(norm), cur_stream()), "gamer_adamw");
    return norm;
}

}  // namespace

TORCH_LIBRARY(gamer, m) {
    m.def("rmsnorm_fwd(Tensor x, Tensor w, float eps, bool bf16_out=False) -> Tensor");
    m.def("rmsnorm_bwd(Tensor x, Tensor w, Tensor dy, float eps) -> (Tensor, Tensor)");
    m.def("linear(Tensor x, Tensor w) -> Tensor");
    m.def("qkv_rope_fwd(Tensor(a!) qkv, int S, int nq, int nkv, Tensor wq, Tensor wk, float eps, Tensor cos, Tensor sin, "
          "Tensor? bias_q=None, Tensor? bias_k=None, Tensor? bias_v=None, Tensor? act_idx=None) -> (Tensor, Tensor)");
    m.def("mb_attention_fwd(Tensor q, Tensor k, Tensor v, Tensor kl, Tensor? ql, Tensor row_empty, Tensor tile_empty, int B, int S, "
          "int nq, int nkv, float scale, float p_drop=0.0, int seed=0) -> (Tensor, Tensor)");
    m.def("mb_attention_bwd(Tensor q, Tensor k, Tensor v, Tensor o, Tensor d_o, Tensor lse, Tensor kl, Tensor? ql, Tensor row_empty, "
          "Tensor tile_empty, int B, int S, int nq, int nkv, float scale, float p_drop=0.0, int seed=0) -> (Tensor, Tensor, Tensor)");
    m.def("swiglu_fwd(Tensor g, Tensor u, float p_drop=0.0, int seed=0) -> Tensor");
    m.def("swiglu_bwd(Tensor g, Tensor u, Tensor dhm, float p_drop=0.0, int seed=0) -> (Tensor, Tensor)");
    m.def("routed_swiglu_fwd(Tensor hin, Tensor w_gate, Tensor w_up, Tensor offsets, float p_drop=0.0, int seed=0) -> (Tensor, Tensor, Tensor)");
    m.def("routed_swiglu_bwd(Tensor g, Tensor u, Tensor dhm, Tensor hin, Tensor w_gate, Tensor w_up, Tensor offsets, float p_drop=0.0, "
          "int seed=0) -> (Tensor, Tensor, Tensor)");
    m.def("lmhead_ce_fwd(Tensor(a!) logits, Tensor labels, int V, float temperature) -> (Tensor, Tensor, Tensor)");
    m.def("lmhead_ce_bwd(Tensor(a!) logits, Tensor labels, Tensor lse, int V, float temperature, Tensor denom, Tensor dloss) -> ()");
    m.def("fused_adamw_clip(Tensor(a!) p, Tensor g, Tensor(b!) m, Tensor(c!) v, int n_decay, float lr, float beta1, float beta2, "
          "float eps, float weight_decay, int step, float max_norm, float grad_scale=1.0) -> Tensor");
}

TORCH_LIBRARY_IMPL(gamer, CUDA, m) {
    m.impl("rmsnorm_fwd", rmsnorm_fwd);
    m.impl("rmsnorm_bwd", rmsnorm_bwd);
    m.impl("linear", linear);
    m.impl("qkv_rope_fwd", qkv_rope_fwd);
    m.impl("mb_attention_fwd", mb_attention_fwd);
    m.impl("mb_attention_bwd", mb_attention_bwd);
    m.impl("swiglu_fwd", swiglu_fwd);
    m.impl("swiglu_bwd", swiglu_bwd);
    m.impl("routed_swiglu_fwd", routed_swiglu_fwd);
    m.impl("routed_swiglu_bwd", routed_swiglu_bwd);
    m.impl("lmhead_ce_fwd", lmhead_ce_fwd);
    m.impl("lmhead_ce_bwd", lmhead_ce_bwd);
    m.impl("fused_adamw_clip", fused_adamw_clip);
}
