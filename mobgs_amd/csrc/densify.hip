// Densification kernels: per-step statistics, selection masks, mask -> row list, and the
// one-launch multi-field row gather behind clone / split / prune of the per-splat table (parameters, Adam moments,
// statistics).  Replaces the torch indexing / cat / repeat sequences of
// /root/reference/scene/gaussian_model.py:1044-1155 (prune_points, cat_tensors_to_optimizer,
// densification_postfix), :1207-1244 (densify_and_splitv2), :1352-1356 (add_densification_stats), :1480-1506
// (densify_and_clone) and /root/reference/helper_train.py:263 (max_radii2D update).
#include "common.h"

namespace mobgs {

// ---- per-step statistics ------------------------------------------------------------------------------------------
// visible[i] != 0 (or, when visible == NULL, radii[i] > 0):
//   max_radii2D[i] = max(max_radii2D[i], radii[i]);  accum[i] += |viewspace_grad[i, :2]|;  denom[i] += 1
__global__ void __launch_bounds__(256)
densify_stats_kernel(int n, const float* __restrict__ vgrad, int vstride, const uint8_t* __restrict__ visible,
                     const int32_t* __restrict__ radii, float* __restrict__ accum, float* __restrict__ denom,
                     float* __restrict__ max_radii) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int r = radii ? radii[i] : 0;
    const bool vis = visible ? visible[i] != 0 : r > 0;
    if (!vis) return;
    if (radii && max_radii) max_radii[i] = fmaxf(max_radii[i], (float)r);
    const float gx = vgrad[(size_t)i * vstride], gy = vgrad[(size_t)i * vstride + 1];
    // sqrt(gx^2 + gy^2) rounded like the separate multiply / add / sqrt of torch.norm (no FMA contraction)
    accum[i] += __fsqrt_rn(__fadd_rn(__fmul_rn(gx, gx), __fmul_rn(gy, gy)));
    denom[i] += 1.f;
}

// ---- selection -------------------------------------------------------------------------------------------------
// mean gradient g = accum / denom (NaN -> 0); big = max_k exp(scaling[i,k]) > size_threshold
//   clone[i] = g >= thr && !big ;  split[i] = g >= thr && big   (rows >= n_grads count as g = 0)
__global__ void __launch_bounds__(256)
densify_select_kernel(int n, int n_grads, const float* __restrict__ accum, const float* __restrict__ denom,
                      const float* __restrict__ scaling, float thr, float size_threshold,
                      uint8_t* __restrict__ clone_sel, uint8_t* __restrict__ split_sel) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float g = 0.f;
    if (i < n_grads) {
        g = accum[i] / denom[i];
        if (g != g) g = 0.f;
    }
    const float s = fmaxf(fmaxf(expf(scaling[3 * i]), expf(scaling[3 * i + 1])), expf(scaling[3 * i + 2]));
    const bool hot = fabsf(g) >= thr;  // torch.norm(grads, dim=-1) of an [N,1] tensor
    const bool hot_split = g >= thr;   // padded_grad >= thr
    const bool big = s > size_threshold;
    clone_sel[i] = (hot && !big) ? 1 : 0;
    split_sel[i] = (hot_split && big) ? 1 : 0;
}

// ---- mask -> ascending list of the rows whose mask byte equals `want` (single workgroup per 2048 rows + look-back
// would be overkill here: the table has a few 100k rows and this runs every 100 iterations; one workgroup walks
// the mask in order) ----------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(1024)
mask_indices_kernel(int n, const uint8_t* __restrict__ mask, int want, int32_t* __restrict__ indices,
                    int32_t* __restrict__ count) {
    __shared__ int wsum[16];
    __shared__ int s_base;
    if (threadIdx.x == 0) s_base = 0;
    __syncthreads();
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    for (int start = 0; start < n; start += 1024) {
        const int i = start + threadIdx.x;
        const bool on = i < n && (mask[i] != 0) == (want != 0);
        const uint64_t ballot = __builtin_amdgcn_ballot_w64(on);
        const int before = __builtin_popcountll(ballot & ((1ull << lane) - 1ull));
        if (lane == 0) wsum[wv] = __builtin_popcountll(ballot);
        __syncthreads();
        int off = s_base, tot = 0;
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            if (k < wv) off += wsum[k];
            tot += wsum[k];
        }
        if (on) indices[off + before] = i;
        __syncthreads();
        if (threadIdx.x == 0) s_base += tot;
        __syncthreads();
    }
    if (threadIdx.x == 0) *count = s_base;
}

// ---- multi-field row gather ---------------------------------------------------------------------------------------
constexpr int MAX_FIELDS = 64;
struct GatherFields {
    const uint8_t* src[MAX_FIELDS];
    uint8_t* dst[MAX_FIELDS];
    int32_t row_bytes[MAX_FIELDS];
    int32_t zero_new[MAX_FIELDS];  // 1: rows whose source index is flagged "new" (index < 0) are written as zeros
};

// dst_f[dst_offset + r] = src_f[|index[r]|-decoded row]  for every field f (blockIdx.y) and output row r.
// index[r] >= 0: copy of row index[r];  index[r] < 0: "new" copy of row -(index[r] + 1) -- fields with zero_new
// (Adam moments, statistics) get zeros instead.
__global__ void __launch_bounds__(256)
rows_gather_kernel(GatherFields F, const int32_t* __restrict__ index, int n_out, int dst_offset) {
    const int f = blockIdx.y;
    const int rb = F.row_bytes[f];
    const uint8_t* __restrict__ src = F.src[f];
    uint8_t* __restrict__ dst = F.dst[f];
    const bool zero_new = F.zero_new[f] != 0;
    if ((rb & 3) == 0) {
        const int words = rb >> 2;
        const size_t total = (size_t)n_out * words;
        for (size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (size_t)gridDim.x * blockDim.x) {
            const int r = (int)(t / words), w = (int)(t - (size_t)r * words);
            const int ix = index[r];
            const int srow = ix >= 0 ? ix : -(ix + 1);
            const uint32_t v = (ix < 0 && zero_new) ? 0u : reinterpret_cast<const uint32_t*>(src)[(size_t)srow * words + w];
            reinterpret_cast<uint32_t*>(dst)[(size_t)(dst_offset + r) * words + w] = v;
        }
    } else {
        const size_t total = (size_t)n_out * rb;
        for (size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (size_t)gridDim.x * blockDim.x) {
            const int r = (int)(t / rb), w = (int)(t - (size_t)r * rb);
            const int ix = index[r];
            const int srow = ix >= 0 ? ix : -(ix + 1);
            dst[(size_t)(dst_offset + r) * rb + w] = (ix < 0 && zero_new) ? (uint8_t)0 : src[(size_t)srow * rb + w];
        }
    }
}

// ---- split children: xyz = R(q_parent) * sample + xyz_parent ; scaling = log(exp(scaling_parent) / (0.8 N)) --------
// rows [first, first + n_children) of xyz / scaling hold copies of the parents (written by the gather); rotation rows
// likewise (children inherit it), so every input is read from the child's own row.
__global__ void __launch_bounds__(256)
split_children_kernel(int n_children, int first, int n_split, const float* __restrict__ samples,
                      const float* __restrict__ rotation, float* __restrict__ xyz, float* __restrict__ scaling) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= n_children) return;
    const size_t row = (size_t)first + c;
    const float* q4 = rotation + 4 * row;
    const float qn = sqrtf(q4[0] * q4[0] + q4[1] * q4[1] + q4[2] * q4[2] + q4[3] * q4[3]);
    const float w = q4[0] / qn, x = q4[1] / qn, y = q4[2] / qn, z = q4[3] / qn;
    const float R[9] = {1.f - 2.f * (y * y + z * z), 2.f * (x * y - w * z), 2.f * (x * z + w * y),
                        2.f * (x * y + w * z), 1.f - 2.f * (x * x + z * z), 2.f * (y * z - w * x),
                        2.f * (x * z - w * y), 2.f * (y * z + w * x), 1.f - 2.f * (x * x + y * y)};
    const float s0 = samples[3 * (size_t)c], s1 = samples[3 * (size_t)c + 1], s2 = samples[3 * (size_t)c + 2];
    float* p = xyz + 3 * row;
#pragma unroll
    for (int k = 0; k < 3; ++k) p[k] = (R[3 * k] * s0 + R[3 * k + 1] * s1 + R[3 * k + 2] * s2) + p[k];
    float* sc = scaling + 3 * row;
    const float inv = 1.f / (0.8f * (float)n_split);
#pragma unroll
    for (int k = 0; k < 3; ++k) sc[k] = logf(expf(sc[k]) * inv);
}

}  // namespace mobgs

// ---- one Adam step over up to 64 tensors (include/mobgs_hip.h K16) -------------------------------------------------
struct AdamTable {
    MobgsAdamTensor t[64];
};
// grid (blocks over the longest tensor, tensor): 16 bytes per lane and stream; tensors shorter than the grid exit at once
__global__ void __launch_bounds__(256) adam_step_kernel(AdamTable tab, float w1, float beta2, float w2, float eps) {
    const MobgsAdamTensor T = tab.t[blockIdx.y];
    const int64_t i0 = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if (i0 >= T.n) return;
    auto one = [&](float& p, float g, float& m, float& v) {
        m = m + w1 * (g - m);                  // exp_avg.lerp_(grad, 1 - beta1)
        v = v * beta2 + w2 * g * g;            // exp_avg_sq.mul_(beta2).addcmul_(grad, grad, value = 1 - beta2)
        const float denom = sqrtf(v) / T.bias2_sqrt + eps;
        p = p - T.step_size * (m / denom);     // param.addcdiv_(exp_avg, denom, value = -step_size)
    };
    if (i0 + 4 <= T.n && ((((uintptr_t)T.param | (uintptr_t)T.grad | (uintptr_t)T.exp_avg | (uintptr_t)T.exp_avg_sq) & 15) == 0)) {
        float4 p = *reinterpret_cast<float4*>(T.param + i0);
        const float4 g = *reinterpret_cast<const float4*>(T.grad + i0);
        float4 m = *reinterpret_cast<float4*>(T.exp_avg + i0);
        float4 v = *reinterpret_cast<float4*>(T.exp_avg_sq + i0);
        one(p.x, g.x, m.x, v.x);
        one(p.y, g.y, m.y, v.y);
        one(p.z, g.z, m.z, v.z);
        one(p.w, g.w, m.w, v.w);
        *reinterpret_cast<float4*>(T.param + i0) = p;
        *reinterpret_cast<float4*>(T.exp_avg + i0) = m;
        *reinterpret_cast<float4*>(T.exp_avg_sq + i0) = v;
    } else {
        for (int64_t i = i0; i < T.n && i < i0 + 4; ++i) one(T.param[i], T.grad[i], T.exp_avg[i], T.exp_avg_sq[i]);
    }
}

using namespace mobgs;

extern "C" {

int mobgs_adam_step(int n_tensors, const MobgsAdamTensor* tensors_host, double beta1, double beta2, double eps,
                    void* stream) {
    if (n_tensors < 0 || n_tensors > 64 || (n_tensors > 0 && !tensors_host)) {
        set_error("mobgs_adam_step: n_tensors = %d (0..64)", n_tensors);
        return MOBGS_E_INVALID;
    }
    if (n_tensors == 0) return MOBGS_OK;
    AdamTable tab;
    int64_t longest = 0;
    for (int i = 0; i < n_tensors; ++i) {
        tab.t[i] = tensors_host[i];
        if (!tab.t[i].param || !tab.t[i].grad || !tab.t[i].exp_avg || !tab.t[i].exp_avg_sq || tab.t[i].n < 0) {
            set_error("mobgs_adam_step: tensor %d has a NULL pointer or a negative size", i);
            return MOBGS_E_INVALID;
        }
        longest = tab.t[i].n > longest ? tab.t[i].n : longest;
    }
    if (longest == 0) return MOBGS_OK;
    const int64_t blocks = (longest + 1023) / 1024;
    hipLaunchKernelGGL(adam_step_kernel, dim3((unsigned)blocks, (unsigned)n_tensors), dim3(256), 0, (hipStream_t)stream,
                       tab, (float)(1.0 - beta1), (float)beta2, (float)(1.0 - beta2), (float)eps);
    return check_launch("adam_step_kernel");
}

int mobgs_densify_stats(int n, const float* viewspace_grad, int grad_stride, const uint8_t* visible,
                        const int32_t* radii, float* xyz_gradient_accum, float* denom, float* max_radii2D,
                        void* stream) {
    if (n < 0 || grad_stride < 2 || (!visible && !radii)) {
        set_error("mobgs_densify_stats: bad arguments n=%d stride=%d", n, grad_stride);
        return MOBGS_E_INVALID;
    }
    if (n == 0) return MOBGS_OK;
    hipLaunchKernelGGL(densify_stats_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, n,
                       viewspace_grad, grad_stride, visible, radii, xyz_gradient_accum, denom, max_radii2D);
    return check_launch("densify_stats_kernel");
}

int mobgs_densify_select(int n, int n_grads, const float* xyz_gradient_accum, const float* denom,
                         const float* scaling, float grad_threshold, float size_threshold, uint8_t* clone_sel,
                         uint8_t* split_sel, void* stream) {
    if (n < 0 || n_grads < 0 || n_grads > n) {
        set_error("mobgs_densify_select: bad sizes n=%d n_grads=%d", n, n_grads);
        return MOBGS_E_INVALID;
    }
    if (n == 0) return MOBGS_OK;
    hipLaunchKernelGGL(densify_select_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, n, n_grads,
                       xyz_gradient_accum, denom, scaling, grad_threshold, size_threshold, clone_sel, split_sel);
    return check_launch("densify_select_kernel");
}

int mobgs_mask_indices(int n, const uint8_t* mask, int want, int32_t* indices, int32_t* count, void* stream) {
    if (n < 0) {
        set_error("mobgs_mask_indices: n=%d", n);
        return MOBGS_E_INVALID;
    }
    hipLaunchKernelGGL(mask_indices_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, n, mask, want, indices, count);
    return check_launch("mask_indices_kernel");
}

int mobgs_rows_gather(int n_fields, const void* const* src_host, void* const* dst_host, const int32_t* row_bytes_host,
                      const int32_t* zero_new_host, const int32_t* index, int n_out, int dst_offset, void* stream) {
    if (n_fields < 0 || n_fields > MAX_FIELDS || n_out < 0 || dst_offset < 0) {
        set_error("mobgs_rows_gather: bad arguments n_fields=%d n_out=%d", n_fields, n_out);
        return MOBGS_E_INVALID;
    }
    if (n_fields == 0 || n_out == 0) return MOBGS_OK;
    GatherFields F;
    int max_rb = 0;
    for (int f = 0; f < n_fields; ++f) {
        F.src[f] = (const uint8_t*)src_host[f];
        F.dst[f] = (uint8_t*)dst_host[f];
        F.row_bytes[f] = row_bytes_host[f];
        F.zero_new[f] = zero_new_host ? zero_new_host[f] : 0;
        if (row_bytes_host[f] < 0 || (row_bytes_host[f] > 0 && (!src_host[f] || !dst_host[f]))) {
            set_error("mobgs_rows_gather: field %d has bad row size or NULL buffers", f);
            return MOBGS_E_INVALID;
        }
        max_rb = max_rb > row_bytes_host[f] ? max_rb : row_bytes_host[f];
    }
    if (max_rb == 0) return MOBGS_OK;
    long long work = ((long long)n_out * ((max_rb + 3) / 4) + 255) / 256;
    const int gx = (int)(work < 1 ? 1 : (work > 4096 ? 4096 : work));
    hipLaunchKernelGGL(rows_gather_kernel, dim3(gx, n_fields), dim3(256), 0, (hipStream_t)stream, F, index, n_out,
                       dst_offset);
    return check_launch("rows_gather_kernel");
}

int mobgs_split_children(int n_children, int first_row, int n_split, const float* samples, const float* rotation,
                         float* xyz, float* scaling, void* stream) {
    if (n_children < 0 || first_row < 0 || n_split < 1) {
        set_error("mobgs_split_children: bad arguments");
        return MOBGS_E_INVALID;
    }
    if (n_children == 0) return MOBGS_OK;
    hipLaunchKernelGGL(split_children_kernel, dim3((n_children + 255) / 256), dim3(256), 0, (hipStream_t)stream,
                       n_children, first_row, n_split, samples, rotation, xyz, scaling);
    return check_launch("split_children_kernel");
}

}  // extern "C"
