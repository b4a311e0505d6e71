// PPO-Lagrangian scalar-side kernels: reward+cost GAE scan, fused SafePPOLogGrad forward+backward,
// value losses, small policy/value heads.  All fp32 (the reference computes these in fp32), HBM/latency-bound.
#include "common.h"

// ------------------------------------------------------------------------------------------------
// GAE reverse scan, reward and cost twins fused.  One thread per env (coalesced over B), sequential in T.
// Restates AllenAct RolloutStorage.compute_returns(use_gae=True) [3P; SURVEY.md App. C]:
//   delta = r[t] + gamma*V[t+1]*m[t+1] - V[t];  g = delta + (gamma*tau)*m[t+1]*g;  ret[t] = g + V[t];  adv = ret - V
// FP contraction is switched off in the kernel body so results are bit-identical to the fp32 torch oracle.
template <int UNROLL>
__global__ void gae_scan_kernel(const float* __restrict__ rewards, const float* __restrict__ costs,
                                const float* __restrict__ values, const float* __restrict__ c_values,
                                const float* __restrict__ masks, const float* __restrict__ next_v,
                                const float* __restrict__ next_cv, float gamma, float gamma_tau, int T, int B,
                                float* __restrict__ ret, float* __restrict__ adv, float* __restrict__ c_ret,
                                float* __restrict__ c_adv) {
#pragma clang fp contract(off)  // forbid FMA contraction: evaluation order/rounding identical to the fp32 torch oracle
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    float g = 0.f, gc = 0.f, vn = next_v[b], cvn = next_cv[b];
    int t = T - 1;
    for (; t >= 0; t -= UNROLL) {
        float r[UNROLL], c[UNROLL], v[UNROLL], cv[UNROLL], m[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {  // issue all loads of the chunk before the dependent chain
            const int tt = t - u;
            const bool ok = tt >= 0;
            const size_t i = (size_t)(ok ? tt : 0) * B + b;
            r[u] = rewards[i]; c[u] = costs[i]; v[u] = values[i]; cv[u] = c_values[i];
            m[u] = masks[i + B];
        }
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            const int tt = t - u;
            if (tt < 0) break;
            const size_t i = (size_t)tt * B + b;
            // plain operators (not the __f*_rn wrappers: those are inline functions outside this pragma's scope)
            const float d = (r[u] + (gamma * vn) * m[u]) - v[u];
            g = d + (gamma_tau * m[u]) * g;
            const float rt = g + v[u];
            ret[i] = rt; adv[i] = rt - v[u];
            vn = v[u];
            const float dc = (c[u] + (gamma * cvn) * m[u]) - cv[u];
            gc = dc + (gamma_tau * m[u]) * gc;
            const float crt = gc + cv[u];
            c_ret[i] = crt; c_adv[i] = crt - cv[u];
            cvn = cv[u];
        }
    }
}

extern "C" int svla_gae_scan_f32(const float* rewards, const float* costs, const float* values, const float* c_values,
                                 const float* masks, const float* next_v, const float* next_cv, double gamma, double tau,
                                 int T, int B, float* ret, float* adv, float* c_ret, float* c_adv, void* stream) {
    if (T <= 0 || B <= 0) return SVLA_EINVAL;
    const int threads = 64;
    hipLaunchKernelGGL(gae_scan_kernel<8>, dim3((B + threads - 1) / threads), dim3(threads), 0, (hipStream_t)stream,
                       rewards, costs, values, c_values, masks, next_v, next_cv, (float)gamma, (float)(gamma * tau), T, B,
                       ret, adv, c_ret, c_adv);
    return svla_launch_status();
}

// ------------------------------------------------------------------------------------------------
// Fused SafePPOLogGrad forward + backward (training/online/loss/customized_loss.py:317-449).
// One thread per (t,b) row, A <= 32 logits in registers.  sums[0..2] += {value_sq_err, action_loss, -entropy}
// (double atomics); dlogits/dvalues are the gradients of
//   total = value_coef*0.5*mean(v_err) + action_w*mean(action_loss) + ent_coef*mean(-H)
// with mean = sum * inv_n (inv_n = 1/rows of the *whole* minibatch, so micro-batches/ranks add up exactly).
template <int A_MAX>
__global__ void ppo_lag_loss_kernel(const float* __restrict__ logits, const float* __restrict__ values,
                                    const int64_t* __restrict__ actions, const float* __restrict__ old_logp,
                                    const float* __restrict__ adv, const float* __restrict__ c_adv,
                                    const float* __restrict__ returns, const float* __restrict__ old_values, int rows,
                                    int A, float lam, float clip, float value_coef, float action_w, float ent_coef,
                                    int use_clipped_value, float inv_n, float* __restrict__ dlogits,
                                    float* __restrict__ dvalues, double* __restrict__ sums) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    float s_v = 0.f, s_a = 0.f, s_e = 0.f;
    if (r < rows) {
        float z[A_MAX];
        float mx = -INFINITY;
#pragma unroll
        for (int j = 0; j < A_MAX; ++j) {
            z[j] = j < A ? logits[(size_t)r * A + j] : -INFINITY;
            mx = fmaxf(mx, z[j]);
        }
        float se = 0.f;
#pragma unroll
        for (int j = 0; j < A_MAX; ++j) se += (j < A) ? expf(z[j] - mx) : 0.f;
        const float lse = mx + logf(se);
        const int a = (int)actions[r];
        float H = 0.f, lp_a = 0.f;
#pragma unroll
        for (int j = 0; j < A_MAX; ++j) {
            if (j < A) {
                const float lp = z[j] - lse;
                const float p = expf(lp);
                H -= p * lp;
                if (j == a) lp_a = lp;
                z[j] = lp;  // keep log-probs
            }
        }
        const float ratio = expf(lp_a - old_logp[r]);
        const float clamped = fminf(fmaxf(ratio, 1.f - clip), 1.f + clip);
        const float advm = (adv[r] - lam * (c_adv ? c_adv[r] : 0.f)) / (1.f + lam);
        const float surr1 = ratio * advm, surr2 = clamped * advm;
        const bool use_clamped = surr2 < surr1;
        s_a = -(use_clamped ? surr2 : surr1);
        const bool inside = (ratio >= 1.f - clip) && (ratio <= 1.f + clip);
        const float dL_dlp = -((use_clamped && !inside) ? 0.f : ratio * advm) * action_w * inv_n;
        s_e = -H;
        const float ce = ent_coef * inv_n;
#pragma unroll
        for (int j = 0; j < A_MAX; ++j) {
            if (j < A) {
                const float p = expf(z[j]);
                dlogits[(size_t)r * A + j] = dL_dlp * ((j == a ? 1.f : 0.f) - p) + ce * p * (z[j] + H);
            }
        }
        const float v = values[r], rt = returns[r];
        if (use_clipped_value) {
            const float ov = old_values[r];
            const float dv = v - ov;
            const float vc = ov + fminf(fmaxf(dv, -clip), clip);
            const float l1 = (v - rt) * (v - rt), l2 = (vc - rt) * (vc - rt);
            s_v = fmaxf(l1, l2);
            // torch.max backward: gradient goes to the larger operand (ties: equal split is measure-zero here)
            float gv = (l1 >= l2) ? 2.f * (v - rt) : ((dv >= -clip && dv <= clip) ? 2.f * (vc - rt) : 0.f);
            dvalues[r] = value_coef * 0.5f * gv * inv_n;
        } else {
            s_v = (rt - v) * (rt - v);
            dvalues[r] = value_coef * (v - rt) * inv_n;
        }
    }
    s_v = wave_sum(s_v); s_a = wave_sum(s_a); s_e = wave_sum(s_e);
    if ((threadIdx.x & 63) == 0) {
        atomicAdd(&sums[0], (double)s_v);
        atomicAdd(&sums[1], (double)s_a);
        atomicAdd(&sums[2], (double)s_e);
    }
}

extern "C" int svla_ppo_lag_loss_fwd_bwd_f32(const float* logits, const float* values, const int64_t* actions,
                                             const float* old_logp, const float* adv, const float* c_adv,
                                             const float* returns, const float* old_values, int rows, int A, float lam,
                                             float clip, float value_coef, float action_w, float ent_coef,
                                             int use_clipped_value, float inv_n, float* dlogits, float* dvalues,
                                             double* sums, void* stream) {
    if (rows <= 0 || A <= 0 || A > 32) return SVLA_EINVAL;
    if (use_clipped_value && !old_values) return SVLA_EINVAL;
    const int threads = 128;
    hipLaunchKernelGGL(ppo_lag_loss_kernel<32>, dim3((rows + threads - 1) / threads), dim3(threads), 0,
                       (hipStream_t)stream, logits, values, actions, old_logp, adv, c_adv, returns, old_values, rows, A, lam,
                       clip, value_coef, action_w, ent_coef, use_clipped_value, inv_n, dlogits, dvalues, sums);
    return svla_launch_status();
}

// PPOValue / SafePPOValue [3P]: 0.5*mean((returns - values)^2); sums[0] += sum sq err; dvalues = coef*(v-ret)*inv_n.
// old_values != NULL: upstream AllenAct's clipped form (use_clipped_value_loss=True, the same expression as
// customized_loss.py:374-380): 0.5*mean(max((v-ret)^2, (clip(v, old +- clip_param) - ret)^2)).
__global__ void value_mse_kernel(const float* __restrict__ values, const float* __restrict__ returns,
                                 const float* __restrict__ old_values, float clip, int rows, float coef, float inv_n,
                                 float* __restrict__ dvalues, double* __restrict__ sums) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    float s = 0.f;
    if (r < rows) {
        const float v = values[r], rt = returns[r];
        const float d = v - rt;
        if (old_values) {
            const float ov = old_values[r], dv = v - ov;
            const float vc = ov + fminf(fmaxf(dv, -clip), clip);
            const float l1 = d * d, l2 = (vc - rt) * (vc - rt);
            s = fmaxf(l1, l2);
            const float gv = (l1 >= l2) ? d : ((dv >= -clip && dv <= clip) ? (vc - rt) : 0.f);
            dvalues[r] = coef * gv * inv_n;
        } else {
            s = d * d;
            dvalues[r] = coef * d * inv_n;
        }
    }
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) atomicAdd(&sums[0], (double)s);
}

extern "C" int svla_value_mse_fwd_bwd_f32(const float* values, const float* returns, const float* old_values, float clip, int rows,
                                          float coef, float inv_n, float* dvalues, double* sums, void* stream) {
    if (rows <= 0) return SVLA_EINVAL;
    hipLaunchKernelGGL(value_mse_kernel, dim3((rows + 127) / 128), dim3(128), 0, (hipStream_t)stream, values, returns, old_values, clip,
                       rows, coef, inv_n, dvalues, sums);
    return svla_launch_status();
}

// ------------------------------------------------------------------------------------------------
// HL-Gauss discrete critic (utils/loss_functions.py:7-30 + DiscreteCriticHead.forward, allenact_dino_transformer.py:743-766), one
// wave per row, forward + backward fused:
//   p = softmax(logits[r, :NB]);   value[r] = sum_j p_j * centre_j,  centre_j = (edge_j + edge_{j+1}) / 2,
//   edge_j = torch.linspace(vmin, vmax, NB + 1)[j] (fp32, hl_edge below);
//   target != NULL:  q_j = (erf((edge_{j+1} - t) / (sqrt(2) sigma)) - erf((edge_j - t) / (sqrt(2) sigma))) / z,
//                    z = erf((edge_NB - t)/..) - erf((edge_0 - t)/..);   loss_r = -sum_j q_j log p_j  (F.cross_entropy with
//                    probability targets);  sums[0] += loss_r;  dlogits = coef * inv_n * (p * sum(q) - q)
//   dvalue != NULL:  dlogits += dvalue[r] * p_j * (centre_j - value[r])     (gradient through transform_from_probs(softmax))
// torch.linspace(vmin, vmax, NB + 1)[j] as PyTorch computes it in fp32 (symmetric: from the start in the lower half, from the end above)
__device__ __forceinline__ float hl_edge(int j, int NB, float vmin, float vmax, float step) {
    return (j < (NB + 1) / 2) ? vmin + step * (float)j : vmax - step * (float)(NB - j);
}

__global__ void hlgauss_kernel(const float* __restrict__ logits, const float* __restrict__ target, const float* __restrict__ dvalue,
                               int rows, int NB, float vmin, float vmax, float sigma, float coef, float inv_n,
                               float* __restrict__ values_out, float* __restrict__ dlogits, double* __restrict__ sums) {
    const int lane = threadIdx.x & 63;
    const int r = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (r >= rows) return;
    const float* x = logits + (size_t)r * NB;
    const float step = (vmax - vmin) / (float)NB;
    float z[4], mx = -INFINITY;             // NB <= 256: up to 4 bins per lane
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int j = lane + 64 * u;
        z[u] = j < NB ? x[j] : -INFINITY;
        mx = fmaxf(mx, z[u]);
    }
    mx = wave_max(mx);
    float se = 0.f;
#pragma unroll
    for (int u = 0; u < 4; ++u) se += (lane + 64 * u < NB) ? expf(z[u] - mx) : 0.f;
    se = wave_sum(se);
    const float lse = mx + logf(se);
    float val = 0.f;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int j = lane + 64 * u;
        if (j < NB) {
            const float e0 = hl_edge(j, NB, vmin, vmax, step), e1 = hl_edge(j + 1, NB, vmin, vmax, step);
            val += expf(z[u] - lse) * (0.5f * (e0 + e1));
        }
    }
    val = wave_sum(val);
    if (values_out && lane == 0) values_out[r] = val;
    float q[4] = {0.f, 0.f, 0.f, 0.f}, qs = 0.f, loss = 0.f;
    if (target) {
        const float t = target[r];
        const float inv = 1.f / (1.41421356237309515f * sigma);
        const float zt = erff((vmax - t) * inv) - erff((vmin - t) * inv);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int j = lane + 64 * u;
            if (j < NB) {
                const float e0 = hl_edge(j, NB, vmin, vmax, step), e1 = hl_edge(j + 1, NB, vmin, vmax, step);
                q[u] = (erff((e1 - t) * inv) - erff((e0 - t) * inv)) / zt;
                qs += q[u];
                loss -= q[u] * (z[u] - lse);
            }
        }
        qs = wave_sum(qs);
        loss = wave_sum(loss);
        if (sums && lane == 0) atomicAdd(&sums[0], (double)loss);
    }
    if (dlogits) {
        const float dv = dvalue ? dvalue[r] : 0.f;
        const float cl = target ? coef * inv_n : 0.f;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int j = lane + 64 * u;
            if (j < NB) {
                const float p = expf(z[u] - lse);
                const float e0 = hl_edge(j, NB, vmin, vmax, step), e1 = hl_edge(j + 1, NB, vmin, vmax, step);
                dlogits[(size_t)r * NB + j] = cl * (p * qs - q[u]) + dv * p * (0.5f * (e0 + e1) - val);
            }
        }
    }
}

extern "C" int svla_hlgauss_fwd_bwd_f32(const float* logits, const float* target, const float* dvalue, int rows, int NB, float vmin,
                                        float vmax, float sigma, float coef, float inv_n, float* values_out, float* dlogits, double* sums,
                                        void* stream) {
    if (rows <= 0 || NB <= 0 || NB > 256 || !(vmax > vmin) || !(sigma > 0.f)) return SVLA_EINVAL;
    hipLaunchKernelGGL(hlgauss_kernel, dim3((rows + 3) / 4), dim3(256), 0, (hipStream_t)stream, logits, target, dvalue, rows, NB, vmin,
                       vmax, sigma, coef, inv_n, values_out, dlogits, sums);
    return svla_launch_status();
}

// ------------------------------------------------------------------------------------------------
// Imitation-learning action loss: nn.CrossEntropyLoss(ignore_index) on [rows, A] fp32 logits, mean over the non-ignored rows
// (/root/reference/architecture/models/transformer_models/early_fusion_tsfm_models.py:93,115-117).  Fused forward + backward,
// one thread per row (A <= 32): sums[0] += sum_r -log_softmax(logits[r])[target[r]];  dlogits = (softmax - onehot) / n_valid.
// ``n_valid`` is read from device memory (counted by the caller without a host sync); rows with target == ignore_index get 0.
__global__ void ce_loss_kernel(const float* __restrict__ logits, const long* __restrict__ target, int rows, int A, long ignore_index,
                               const float* __restrict__ n_valid, float* __restrict__ dlogits, double* __restrict__ sums) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    float loss = 0.f;
    if (r < rows) {
        const float* x = logits + (size_t)r * A;
        float* dx = dlogits + (size_t)r * A;
        const long t = target[r];
        if (t == ignore_index || t < 0 || t >= A) {
            for (int a = 0; a < A; ++a) dx[a] = 0.f;
        } else {
            float mx = x[0];
            for (int a = 1; a < A; ++a) mx = fmaxf(mx, x[a]);
            float se = 0.f;
            for (int a = 0; a < A; ++a) se += __expf(x[a] - mx);
            const float lse = mx + __logf(se);
            const float inv = 1.f / fmaxf(*n_valid, 1.f);
            for (int a = 0; a < A; ++a) dx[a] = (__expf(x[a] - lse) - (a == (int)t ? 1.f : 0.f)) * inv;
            loss = lse - x[t];
        }
    }
    loss = wave_sum(loss);
    if ((threadIdx.x & 63) == 0) atomicAdd(&sums[0], (double)loss);
}

extern "C" int svla_ce_loss_fwd_bwd_f32(const float* logits, const long* target, int rows, int A, long ignore_index,
                                        const float* n_valid, float* dlogits, double* sums, void* stream) {
    if (rows <= 0 || A <= 0 || A > 1024 || !n_valid) return SVLA_EINVAL;
    hipLaunchKernelGGL(ce_loss_kernel, dim3((rows + 127) / 128), dim3(128), 0, (hipStream_t)stream, logits, target, rows, A,
                       ignore_index, n_valid, dlogits, sums);
    return svla_launch_status();
}

// ------------------------------------------------------------------------------------------------
// Small heads on fp32 beliefs (AllenAct LinearActorHead / LinearCriticHead [3P]): out[r, n] = x[r,:].W[n,:] + b[n],
// D <= 1024 in 512-wide slices (8 values per lane and slice, one wave per row; the policy's D = 512 is one slice), N <= 32.  ``row_perm_T``/``row_perm_B`` > 0:
// x rows are stored (b*T + t) (decoder layout) while out rows are (t*B + b) (the [step, sampler] layout of the API).
template <int N_MAX>
__device__ __forceinline__ void small_linear_fwd_kernel_body(const float* __restrict__ x, const float* __restrict__ W,
                                        const float* __restrict__ bias, int rows, int N, int D, int T, int B,
                                        float* __restrict__ out) {
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63;
    if (wave >= rows) return;
    int src = wave;
    if (T > 0) { const int t = wave / B, b = wave % B; src = b * T + t; }
    float4 xv[2][2];
#pragma unroll
    for (int ch = 0; ch < 2; ++ch) {
        const int c = ch * 512 + lane * 8;
        const float4* xp = (const float4*)(x + (size_t)src * D + c);
        xv[ch][0] = c < D ? xp[0] : float4{0.f, 0.f, 0.f, 0.f};
        xv[ch][1] = c < D ? xp[1] : float4{0.f, 0.f, 0.f, 0.f};
    }
    for (int n = 0; n < N; ++n) {
        float s = 0.f;
#pragma unroll
        for (int ch = 0; ch < 2; ++ch) {
            const int c = ch * 512 + lane * 8;
            if (c < D) {
                const float4* wp = (const float4*)(W + (size_t)n * D + c);
                const float4 w0 = wp[0], w1 = wp[1], x0 = xv[ch][0], x1 = xv[ch][1];
                // (D = 512: the same expression, in the same order, as the one-slice kernel of earlier rounds)
                const float p = x0.x * w0.x + x0.y * w0.y + x0.z * w0.z + x0.w * w0.w + x1.x * w1.x + x1.y * w1.y + x1.z * w1.z + x1.w * w1.w;
                s = ch == 0 ? p : s + p;
            }
        }
        s = wave_sum(s);
        if (lane == 0) out[(size_t)wave * N + n] = s + (bias ? bias[n] : 0.f);
    }
}
template <int N_MAX>
__global__ void small_linear_fwd_kernel(const float* __restrict__ x, const float* __restrict__ W,
                                        const float* __restrict__ bias, int rows, int N, int D, int T, int B,
                                        float* __restrict__ out) { small_linear_fwd_kernel_body<N_MAX>(x, W, bias, rows, N, D, T, B, out); }

// dx[src_row,:] (+)= sum_n dout[r,n] W[n,:]; dW[n,:] += sum_r dout[r,n] x[src_row,:]; db[n] += sum_r dout[r,n]
template <int N_MAX>
__global__ void small_linear_bwd_kernel(const float* __restrict__ x, const float* __restrict__ W,
                                        const float* __restrict__ dout, int rows, int N, int D, int T, int B, int accumulate_dx,
                                        float* __restrict__ dx, float* __restrict__ dW, float* __restrict__ db, DetCfg det) {
    const int lane = threadIdx.x & 63;
    const int c0 = blockIdx.y * 512 + lane * 8;         // this block's 512-wide column slice (D = 512: the only one)
    if (c0 >= D) return;
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int nwaves = (gridDim.x * blockDim.x) >> 6;
    float gw[N_MAX][8];
    float gb[N_MAX];
#pragma unroll
    for (int n = 0; n < N_MAX; ++n) {
        gb[n] = 0.f;
#pragma unroll
        for (int k = 0; k < 8; ++k) gw[n][k] = 0.f;
    }
    for (int r = wave; r < rows; r += nwaves) {
        int src = r;
        if (T > 0) { const int t = r / B, b = r % B; src = b * T + t; }
        const float4* xp = (const float4*)(x + (size_t)src * D + c0);
        const float4 x0 = xp[0], x1 = xp[1];
        const float xv[8] = {x0.x, x0.y, x0.z, x0.w, x1.x, x1.y, x1.z, x1.w};
        float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
        for (int n = 0; n < N_MAX; ++n) {
            if (n < N) {
                const float g = dout[(size_t)r * N + n];
                const float4* wp = (const float4*)(W + (size_t)n * D + c0);
                const float4 w0 = wp[0], w1 = wp[1];
                const float wv[8] = {w0.x, w0.y, w0.z, w0.w, w1.x, w1.y, w1.z, w1.w};
#pragma unroll
                for (int k = 0; k < 8; ++k) { acc[k] += g * wv[k]; gw[n][k] += g * xv[k]; }
                gb[n] += g;
            }
        }
        float* dp = dx + (size_t)src * D + c0;
#pragma unroll
        for (int k = 0; k < 8; ++k) dp[k] = accumulate_dx ? dp[k] + acc[k] : acc[k];
    }
#pragma unroll
    for (int n = 0; n < N_MAX; ++n) {
        if (n < N) {
#pragma unroll
            for (int k = 0; k < 8; ++k) grad_add(det, &dW[(size_t)n * D + c0 + k], gw[n][k]);
            if (lane == 0 && blockIdx.y == 0 && db) grad_add(det, &db[n], gb[n]);
        }
    }
}

extern "C" int svla_small_linear_fwd_f32(const float* x, const float* W, const float* bias, int rows, int N, int D, int T,
                                         int B, float* out, void* stream) {
    if (D <= 0 || D > 1024 || (D % 8) || N <= 0 || N > 32 || rows <= 0) return SVLA_EINVAL;
    if (T > 0 && T * B != rows) return SVLA_EINVAL;
    SVLA_LAUNCH(small_linear_fwd_kernel<32>, small_linear_fwd_kernel_body<32>, 1024, 1, dim3((rows + 3) / 4), dim3(256), 0, (hipStream_t)stream, x, W, bias,
                       rows, N, D, T, B, out);
    return svla_launch_status();
}

extern "C" int svla_small_linear_bwd_f32(const float* x, const float* W, const float* dout, int rows, int N, int D, int T,
                                         int B, int accumulate_dx, float* dx, float* dW, float* db, void* stream) {
    if (D <= 0 || D > 1024 || (D % 8) || N <= 0 || N > 20 || rows <= 0) return SVLA_EINVAL;
    if (T > 0 && T * B != rows) return SVLA_EINVAL;
    int blocks = (rows + 3) / 4;
    if (blocks > 256) blocks = 256;
    const dim3 grid(blocks, (D + 511) / 512);
    if (N <= 1)
        hipLaunchKernelGGL(small_linear_bwd_kernel<1>, grid, dim3(256), 0, (hipStream_t)stream, x, W, dout, rows, N, D,
                           T, B, accumulate_dx, dx, dW, db, g_svla_det);
    else
        hipLaunchKernelGGL(small_linear_bwd_kernel<20>, grid, dim3(256), 0, (hipStream_t)stream, x, W, dout, rows,
                           N, D, T, B, accumulate_dx, dx, dW, db, g_svla_det);
    return svla_launch_status();
}
