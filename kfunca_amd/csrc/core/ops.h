// namespace gpu: the operator API (reference: src/core/include/*_ops.h, tensor_shape.h).
// Each op builds a TensorIterator (or validates shapes), asks the allocator for outputs/scratch,
// and crosses the C ABI exactly once per kernel.
#pragma once

#include <tuple>
#include <vector>

#include "tensor.h"

namespace gpu {

// binary_ops.h:5-18
Tensor &add_out(Tensor &out, const Tensor &left, const Tensor &right);
Tensor add(const Tensor &left, const Tensor &right);
Tensor &add_(Tensor &self, const Tensor &other);
Tensor &sub_out(Tensor &out, const Tensor &left, const Tensor &right);
Tensor sub(const Tensor &left, const Tensor &right);
Tensor &sub_(Tensor &self, const Tensor &other);
Tensor &mul_out(Tensor &out, const Tensor &left, const Tensor &right);
Tensor mul(const Tensor &left, const Tensor &right);
Tensor &mul_(Tensor &self, const Tensor &other);
Tensor &div_out(Tensor &out, const Tensor &left, const Tensor &right);
Tensor div(const Tensor &left, const Tensor &right);
Tensor &div_(Tensor &self, const Tensor &other);
// tensor (op) python-float, register.cpp:172-206 — same results as the reference's fill-then-op, no temporary
Tensor add(const Tensor &self, double s);
Tensor sub(const Tensor &self, double s);
Tensor mul(const Tensor &self, double s);
Tensor div(const Tensor &self, double s);
Tensor &add_(Tensor &self, double s);
Tensor &sub_(Tensor &self, double s);
Tensor &mul_(Tensor &self, double s);
Tensor &div_(Tensor &self, double s);
// unary_ops.h:5-9, nullary_ops.h:5-8
Tensor clone(const Tensor &self);
Tensor &copy_(Tensor &self, const Tensor &other);
Tensor convert(const Tensor &self, ScalarType dtype);
Tensor &fill_out(Tensor &out, const any_t &value);
Tensor &fill_(Tensor &self, const any_t &value);
// reduce_ops.h:5-9
Tensor sum(const Tensor &self, int64_t reduce_dim);
Tensor mean(const Tensor &self, int64_t reduce_dim);
std::tuple<Tensor, Tensor> mean_var(const Tensor &self, int64_t reduce_dim, bool take_sqrt);
// gemm_ops.h:5-8, nn_ops.h:5-7
void gemm_out(Tensor &out, const Tensor &a, const Tensor &b, float alpha, float beta);
Tensor gemm(const Tensor &a, const Tensor &b, float alpha, float beta);
Tensor causal_attention(const Tensor &q, const Tensor &k, const Tensor &v);
// index_ops.h:5-9, tensor_shape.h:5-10
Tensor &index_put_(Tensor &self, const std::vector<Tensor> &indices, const Tensor &values);
Tensor concat(const std::vector<Tensor> tensors, int64_t dim);
std::vector<Tensor> tensor_split(const Tensor &self, std::vector<int64_t> indices, int64_t dim);
// norm_ops.h, sort_ops.h
std::tuple<Tensor, Tensor> norm_stat(const Tensor &self, int64_t dim);
std::tuple<Tensor, Tensor> sort(const Tensor &self, int64_t dim, bool descending);
std::tuple<Tensor, Tensor> topk(const Tensor &self, int64_t k, int64_t dim, bool largest);

// The reference's roadmap operators (README.md:28-32), finished on its building blocks; all carry autograd.
//   rms_norm / layer_norm: normalise the LAST dim (weight / bias: 1-D of that length, may be undefined), statistics in f32
//   embedding: out[..., :] = table[indices[...], :] (indices Long, negative ones wrap; table 2-D)
Tensor rms_norm(const Tensor &x, const Tensor &weight, double eps);
Tensor layer_norm(const Tensor &x, const Tensor &weight, const Tensor &bias, double eps);
Tensor embedding(const Tensor &table, const Tensor &indices);
//   causal_attention_qkv (the roadmap's qkv_linear, README.md:32, attention side): qkv is the packed output [B*S, 3*H*D] of the
//   QKV projection (columns q | k | v, each H heads of D); q, k, v are read IN PLACE, the result is [B*S, H*D] - the layout
//   the output projection takes - and the backward writes one packed gradient: no split / permute / contiguous copies
Tensor causal_attention_qkv(const Tensor &qkv, int64_t B, int64_t S, int64_t H);
//   gemm_fused (fused projections, README.md:32): out = (alpha a b + bias) o mul + add in ONE kernel (kf_gemm_ex) - what the
//   reference API spells gemm + add + mul + add over three extra passes of the output. bias [N]; mul, add shaped like the output;
//   any of the three may be undefined. Residual connection: add = the stream; gated MLP: mul = the other projection.
Tensor gemm_fused(const Tensor &a, const Tensor &b, float alpha, const Tensor &bias, const Tensor &mul, const Tensor &add);

// extensions used by the backward passes (no reference counterpart)
Tensor gemm_ex(const Tensor &a, bool trans_a, const Tensor &b, bool trans_b, float alpha);
std::tuple<Tensor, Tensor> causal_attention_fwd(const Tensor &q, const Tensor &k, const Tensor &v); // (out, lse)
std::tuple<Tensor, Tensor, Tensor> causal_attention_bwd(const Tensor &q, const Tensor &k, const Tensor &v, const Tensor &out,
                                                        const Tensor &lse, const Tensor &grad_out);

} // namespace gpu
