// Native orchestration of the first half of a rasterization: projection -> intersection offsets (+ reach test)
// -> [one 24-byte read-back] -> emit -> per-tile depth sort, as ONE C call.
//
// The stages are the same entry points a caller can drive one by one (mobgs_project_fwd, mobgs_isect_offsets,
// mobgs_isect_emit_sort); doing it here removes the host gaps a Python driver leaves between ~12 short kernels
// (allocation + ctypes + launch, 10-40 us each while the GPU idles -- profiles/r01: 0.2 ms of a 1.9 ms step).
// Buffers whose size depends on the intersection count live in a caller-owned arena sized from the previous call;
// when it is too small the function returns MOBGS_E_CAPACITY with the required sizes in `stats_host` and has
// written nothing past the arena.
#include "common.h"

using namespace mobgs;

extern "C" {

int mobgs_project_and_bin(int C, int N, const float* means, const float* quats, const float* scales,
                          const float* viewmats, const float* Ks, const float* opacities, int opac_per_camera,
                          int width, int height, float eps2d, float near_plane, float far_plane, float radius_clip,
                          int cull, int32_t* radii, float* means2d, float* depths, float* conics,
                          int32_t* tiles_per_gauss, int32_t* cum_tiles, int32_t* tile_offsets, int32_t* tile_order,
                          int64_t* stats_dev,
                          int capacity_box, int32_t* keep_scan, void* scratch, int64_t capacity_listed,
                          int32_t* flatten_ids, uint64_t* sort_keys, uint64_t* isect_ids, int64_t* stats_host,
                          const MobgsTuning* tuning, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    const int tile_w = (width + MOBGS_TILE - 1) / MOBGS_TILE, tile_h = (height + MOBGS_TILE - 1) / MOBGS_TILE;
    int rc = mobgs::project_fwd_launch(C, N, means, quats, scales, viewmats, Ks, width, height, eps2d, near_plane, far_plane,
                               radius_clip, radii, means2d, depths, conics, tiles_per_gauss, nullptr, 0, PackArgs{nullptr, nullptr, nullptr, 0, 0, 0, 0}, stream,
                                       tuning_geometry_per_camera(tuning));
    if (rc != MOBGS_OK) return rc;
    rc = mobgs_isect_offsets(C, N, tile_w, tile_h, width, height, cull, capacity_box, tiles_per_gauss, means2d, radii,
                             conics, opacities, opac_per_camera, cum_tiles, keep_scan, tile_offsets, tile_order, /*capacity_listed (checked on the host)*/ 0,
                             stats_dev, scratch, tuning,
                             stream);
    if (rc != MOBGS_OK) return rc;
    // the pipeline's one host synchronisation (upstream gsplat has the same one): {I_box, I_listed, longest list}
    // (busy-polling hipStreamQuery instead of a blocking hipStreamSynchronize: the wait is ~0.2 ms at most and a
    // blocking wait adds ~30 us of wake-up latency during which the GPU idles)
    hipError_t e = hipMemcpyAsync(stats_host, stats_dev, 3 * sizeof(int64_t), hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) {
        while ((e = hipStreamQuery(st)) == hipErrorNotReady) {
        }
    }
    if (e != hipSuccess) {
        set_error("mobgs_project_and_bin: statistics read-back failed: %s", hipGetErrorString(e));
        return MOBGS_E_LAUNCH;
    }
    if (stats_host[0] > (int64_t)capacity_box || stats_host[1] > capacity_listed) {
        set_error("mobgs_project_and_bin: arena too small (box %lld > %d or listed %lld > %lld)",
                  (long long)stats_host[0], capacity_box, (long long)stats_host[1], (long long)capacity_listed);
        return MOBGS_E_CAPACITY;
    }
    return mobgs_isect_emit_sort(C, N, tile_w, tile_h, capacity_box, stats_host[1], stats_host[2], depths, cum_tiles,
                                 tile_offsets, scratch, sort_keys, flatten_ids, isect_ids, stream);
}

// seg_stride > 0: the fused single-pass lists (isect.hip, isect_fused_launch) -- sort_keys is then the strided key
// arena [C * n_tiles][8][seg_stride]; 0: the two-pass path
static int project_and_bin_enqueue(int C, int N, const float* means, const float* quats, const float* scales,
                                   const float* viewmats, const float* Ks, const float* opacities,
                                   int opac_per_camera, int width, int height, float eps2d, float near_plane,
                                   float far_plane, float radius_clip, int cull, int32_t* radii, float* means2d,
                                   float* depths, float* conics, int32_t* tiles_per_gauss, int32_t* cum_tiles,
                                   int32_t* tile_offsets, int32_t* tile_order, int64_t* stats_dev,
                                   int capacity_box, int32_t* keep_scan, void* scratch, int64_t capacity_listed,
                                   int32_t* flatten_ids, uint64_t* sort_keys, int seg_stride,
                                   const int32_t* enum_order, uint64_t* isect_ids,
                                   int64_t max_tile_len_hint, int64_t* stats_host_pinned, int64_t stats_seq,
                                   const float* pack_colors, int colors_per_camera, int pack_channels,
                                   float* pack_records, const MobgsTuning* tuning, void* stream,
                                   const MobgsPrepInputs* prep = nullptr) {
    hipStream_t st = (hipStream_t)stream;
    const int tile_w = (width + MOBGS_TILE - 1) / MOBGS_TILE, tile_h = (height + MOBGS_TILE - 1) / MOBGS_TILE;
    if (!stats_host_pinned || capacity_listed < 1) {
        set_error("mobgs_project_and_bin_speculative: stats_host_pinned and capacity_listed are required");
        return MOBGS_E_INVALID;
    }
    // two launches fewer on the critical path: project_fwd clears the binning counters on the way, and tile_scan
    // writes the host's copy of the counts itself when the pinned slot is mapped into the device address space
    int32_t* zero_ptr = nullptr;
    size_t zero_n = 0;
    const long long n_all = (long long)C * N, nt_all = (long long)C * tile_w * tile_h;
    const bool fuse_zero = N > 0 && capacity_box >= 1 && n_all < (1ll << 31) - 1 && nt_all < (1ll << 31) - 1 &&
                           ((uintptr_t)scratch & 7) == 0;
    if (fuse_zero) mobgs::isect_zeroed_region(scratch, (size_t)n_all, (size_t)nt_all, (size_t)capacity_box, &zero_ptr, &zero_n);
    PackArgs pack{nullptr, nullptr, nullptr, 0, 0, 0, 0};
    if (pack_records) {
        if ((!pack_colors && !prep) || !opacities || pack_channels < 0) {
            set_error("mobgs_project_and_bin_speculative: pack_records needs pack_colors and opacities");
            return MOBGS_E_INVALID;
        }
        pack = PackArgs{opacities, pack_colors, pack_records, opac_per_camera, colors_per_camera, pack_channels,
                        mobgs_record_stride(pack_channels + 1)};
    }
    BinArgs bin{nullptr, nullptr, 0, 0};
    if (seg_stride > 0) {
        if (!fuse_zero || !opacities || (long long)capacity_box < 4 * n_all + 2 || ((uintptr_t)scratch & 127) != 0) {
            set_error("mobgs_project_and_bin_fused: needs N > 0, opacities, a 128-byte aligned scratch and capacity_box >= 4 C N + 2");
            return MOBGS_E_INVALID;
        }
        bin = BinArgs{mobgs::isect_bin_records(scratch, (size_t)n_all, (size_t)nt_all, (size_t)capacity_box), opacities,
                      opac_per_camera, cull};
    }
    int rc = mobgs::project_fwd_launch(C, N, means, quats, scales, viewmats, Ks, width, height, eps2d, near_plane, far_plane,
                                       radius_clip, radii, means2d, depths, conics, tiles_per_gauss, zero_ptr, zero_n, pack,
                                       stream, tuning_geometry_per_camera(tuning), bin, prep);
    if (rc != MOBGS_OK) return rc;
    // the binning variant follows the caller's expectation of the longest list (max_tile_len_hint)
    MobgsTuning tn = tuning ? *tuning : MobgsTuning{-1, -1, -1, -1, -1, 0, -1, 0, 0};
    tn.longest_list_hint = (int32_t)(max_tile_len_hint > 0x7fffffff ? 0x7fffffff : max_tile_len_hint);
    void* mirror = nullptr;
    if (hipHostGetDevicePointer(&mirror, stats_host_pinned, 0) != hipSuccess) {
        (void)hipGetLastError();
        mirror = nullptr;
    }
    if (seg_stride > 0)
        rc = mobgs::isect_fused_launch(C, N, tile_w, tile_h, width, height, capacity_box, tiles_per_gauss, cum_tiles,
                                       keep_scan, tile_offsets, tile_order, capacity_listed, stats_dev, scratch,
                                       (int64_t*)mirror, mirror ? stats_seq : 0, sort_keys, seg_stride, flatten_ids,
                                       isect_ids, max_tile_len_hint, enum_order, &tn, stream);
    else
        rc = mobgs::isect_offsets_launch(C, N, tile_w, tile_h, width, height, cull, capacity_box, tiles_per_gauss, means2d,
                                         radii, conics, opacities, opac_per_camera, cum_tiles, keep_scan, tile_offsets,
                                         tile_order, capacity_listed, stats_dev, scratch, fuse_zero, (int64_t*)mirror,
                                         mirror ? stats_seq : 0, &tn, stream);
    if (rc != MOBGS_OK) return rc;
    if (!mirror) {
        hipError_t e = hipMemcpyAsync(stats_host_pinned, stats_dev, 3 * sizeof(int64_t), hipMemcpyDeviceToHost, st);
        if (e != hipSuccess) {
            set_error("mobgs_project_and_bin_speculative: statistics copy failed: %s", hipGetErrorString(e));
            return MOBGS_E_LAUNCH;
        }
    }
    if (seg_stride == 0) {
        rc = mobgs_isect_emit_sort_speculative(C, N, tile_w, tile_h, capacity_box, capacity_listed, max_tile_len_hint,
                                               depths, cum_tiles, tile_offsets, stats_dev, scratch, sort_keys,
                                               flatten_ids, isect_ids, stream);
        if (rc != MOBGS_OK) return rc;
    }
    // 1: the counts travel by an ordinary asynchronous copy (or no sequence number was asked for) -- the caller
    // records an event behind this call and waits on it; 0: poll stats_host_pinned[3] for stats_seq instead
    return (mirror && stats_seq) ? MOBGS_OK : 1;
}

int mobgs_project_and_bin_speculative(int C, int N, const float* means, const float* quats, const float* scales,
                                      const float* viewmats, const float* Ks, const float* opacities,
                                      int opac_per_camera, int width, int height, float eps2d, float near_plane,
                                      float far_plane, float radius_clip, int cull, int32_t* radii, float* means2d,
                                      float* depths, float* conics, int32_t* tiles_per_gauss, int32_t* cum_tiles,
                                      int32_t* tile_offsets, int32_t* tile_order, int64_t* stats_dev,
                                      int capacity_box, int32_t* keep_scan, void* scratch, int64_t capacity_listed,
                                      int32_t* flatten_ids, uint64_t* sort_keys, uint64_t* isect_ids,
                                      int64_t max_tile_len_hint, int64_t* stats_host_pinned, int64_t stats_seq,
                                      const float* pack_colors, int colors_per_camera, int pack_channels,
                                      float* pack_records, const MobgsTuning* tuning, void* stream) {
    return project_and_bin_enqueue(C, N, means, quats, scales, viewmats, Ks, opacities, opac_per_camera, width, height,
                                   eps2d, near_plane, far_plane, radius_clip, cull, radii, means2d, depths, conics,
                                   tiles_per_gauss, cum_tiles, tile_offsets, tile_order, stats_dev, capacity_box,
                                   keep_scan, scratch, capacity_listed, flatten_ids, sort_keys, 0, nullptr, isect_ids,
                                   max_tile_len_hint, stats_host_pinned, stats_seq, pack_colors, colors_per_camera,
                                   pack_channels, pack_records, tuning, stream);
}

int mobgs_project_and_bin_fused(int C, int N, const float* means, const float* quats, const float* scales,
                                const float* viewmats, const float* Ks, const float* opacities, int opac_per_camera,
                                int width, int height, float eps2d, float near_plane, float far_plane,
                                float radius_clip, int cull, int32_t* radii, float* means2d, float* depths,
                                float* conics, int32_t* tiles_per_gauss, int32_t* cum_tiles, int32_t* tile_offsets,
                                int32_t* tile_order, int64_t* stats_dev, int capacity_box, int32_t* keep_scan,
                                void* scratch, int64_t capacity_listed, int32_t* flatten_ids, uint64_t* seg_keys,
                                int seg_stride, const int32_t* enum_order, uint64_t* isect_ids, int64_t max_tile_len_hint,
                                int64_t* stats_host_pinned, int64_t stats_seq, const float* pack_colors,
                                int colors_per_camera, int pack_channels, float* pack_records,
                                const MobgsTuning* tuning, void* stream) {
    if (seg_stride < 1 || !seg_keys) {
        set_error("mobgs_project_and_bin_fused: seg_stride >= 1 and seg_keys are required");
        return MOBGS_E_INVALID;
    }
    return project_and_bin_enqueue(C, N, means, quats, scales, viewmats, Ks, opacities, opac_per_camera, width, height,
                                   eps2d, near_plane, far_plane, radius_clip, cull, radii, means2d, depths, conics,
                                   tiles_per_gauss, cum_tiles, tile_offsets, tile_order, stats_dev, capacity_box,
                                   keep_scan, scratch, capacity_listed, flatten_ids, seg_keys, seg_stride, enum_order,
                                   isect_ids, max_tile_len_hint, stats_host_pinned, stats_seq, pack_colors, colors_per_camera,
                                   pack_channels, pack_records, tuning, stream);
}

int mobgs_prep_project_and_bin_fused(const MobgsPrepInputs* prep, float* means, float* quats, float* scales,
                                     const float* viewmats, const float* Ks, float* opacities, int width, int height,
                                     float eps2d, float near_plane, float far_plane, float radius_clip, int cull,
                                     int32_t* radii, float* means2d, float* depths, float* conics,
                                     int32_t* tiles_per_gauss, int32_t* cum_tiles, int32_t* tile_offsets,
                                     int32_t* tile_order, int64_t* stats_dev, int capacity_box, int32_t* keep_scan,
                                     void* scratch, int64_t capacity_listed, int32_t* flatten_ids, uint64_t* seg_keys,
                                     int seg_stride, const int32_t* enum_order, uint64_t* isect_ids,
                                     int64_t max_tile_len_hint, int64_t* stats_host_pinned, int64_t stats_seq,
                                     float* pack_records, const MobgsTuning* tuning, void* stream) {
    if (!prep || !means || !quats || !scales || !opacities || !pack_records) {
        set_error("mobgs_prep_project_and_bin_fused: prep, the four state outputs and pack_records are required");
        return MOBGS_E_INVALID;
    }
    if (prep->Ns < 0 || prep->Nd < 0 || prep->Ns + prep->Nd < 1) {
        set_error("mobgs_prep_project_and_bin_fused: bad sizes Ns=%d Nd=%d", prep->Ns, prep->Nd);
        return MOBGS_E_INVALID;
    }
    // seg_stride 0: the two-pass lists (first frame of a workload, or the caller's choice)
    return project_and_bin_enqueue(1, prep->Ns + prep->Nd, means, quats, scales, viewmats, Ks, opacities, 0, width, height,
                                   eps2d, near_plane, far_plane, radius_clip, cull, radii, means2d, depths, conics,
                                   tiles_per_gauss, cum_tiles, tile_offsets, tile_order, stats_dev, capacity_box,
                                   keep_scan, scratch, capacity_listed, flatten_ids, seg_keys, seg_stride, enum_order,
                                   isect_ids, max_tile_len_hint, stats_host_pinned, stats_seq, nullptr, 0, 9, pack_records,
                                   tuning, stream, prep);
}

}  // extern "C"
