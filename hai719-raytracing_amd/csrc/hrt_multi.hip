// hrt_render_multi -- the whole frame on several GPUs from ONE process (include/hrt.h), the multi-GPU form of the
// reference's single caller ray_trace_from_camera() (main.cpp:200-263; its own fan-out is one std::thread per scanline,
// :232-238).  Included by hrt_api.hip inside its extern "C" block.
//
//   replicas   the scene is small (<= ~25 MB) and read-only: one hrt_scene per slot, on that slot's device
//   partition  8x8 tiles round-robin, slot r renders tiles r, r + N, ... (hrt_render_tiles), all slots at once, each on a
//              stream of its own; per-pixel RNG keys make the partition invisible in the pixels
//   gather     ONE step: every slot's dense tile buffer is copied device-to-device (hipMemcpyPeerAsync: xGMI DMA between
//              GPUs, a plain copy when a slot shares slot 0's device) into its block of slot 0's gather buffer, on the
//              slot's own stream right behind its kernel; slot 0 renders straight into its block.  No reduction: slots
//              own disjoint pixels.  (One process per GPU with torch.distributed -- bench.py -- uses RCCL's gather for
//              the same step; in one process the runtime's peer copy is the same transfer without a communicator, and
//              it also accepts a repeated ordinal, which a communicator does not.)
//   assemble   slot 0: tiles -> row-major frame (hrt_assemble_kernel), one D2H copy
struct hrt_multi {
    uint32_t n = 0;
    std::vector<int> ordinal;
    std::vector<hrt_scene *> replica;
    std::vector<hipStream_t> stream;
    std::vector<hipEvent_t> done;
    std::vector<float *> d_tiles;      // slots 1..n-1: this slot's tiles on its own device
    size_t tiles_cap = 0;              // floats per slot buffer
    float *d_gathered = nullptr;       // slot 0's device: n blocks of tiles_cap floats
    float *d_frame = nullptr;
    size_t frame_cap = 0;
};

static void multi_free_buffers(hrt_multi *m) {
    if (!m->n || !m->replica[0]) return;  // creation failed before any buffer existed
    for (uint32_t i = 0; i < m->n; ++i)
        if (i < m->d_tiles.size() && m->d_tiles[i]) { (void)hipSetDevice(m->ordinal[i]); (void)hipFree(m->d_tiles[i]); m->d_tiles[i] = nullptr; }
    if (m->n) (void)hipSetDevice(m->ordinal[0]);
    if (m->d_gathered) (void)hipFree(m->d_gathered);
    if (m->d_frame) (void)hipFree(m->d_frame);
    m->d_gathered = nullptr; m->d_frame = nullptr; m->tiles_cap = 0; m->frame_cap = 0;
}

void hrt_multi_destroy(hrt_multi *m) {
    if (!m) return;
    multi_free_buffers(m);
    for (uint32_t i = 0; i < m->n; ++i) {
        if (!m->replica[i]) continue;  // a slot hrt_multi_create never reached (e.g. a bad ordinal): nothing on it
        (void)hipSetDevice(m->ordinal[i]);
        if (i < m->stream.size() && m->stream[i]) (void)hipStreamDestroy(m->stream[i]);
        if (i < m->done.size() && m->done[i]) (void)hipEventDestroy(m->done[i]);
        if (i < m->replica.size() && m->replica[i]) hrt_scene_destroy(m->replica[i]);
    }
    if (m->n && g_rt.ready && m->replica[0]) (void)use_device(m->ordinal[0]);
    (void)hipGetLastError();  // nothing above may leave a sticky error for the next launch check
    delete m;
}

int hrt_multi_create(const hrt_scene_desc *desc, uint32_t n_devices, const int *device_ordinals, hrt_multi **out) {
    if (!desc || !out || !n_devices || !device_ordinals) return fail(HRT_ERR_INVALID, "hrt_multi_create: bad argument");
    if (n_devices > 64u) return fail(HRT_ERR_INVALID, "hrt_multi_create: more than 64 slots");
    hrt_multi *m = new hrt_multi();
    m->n = n_devices;
    m->ordinal.assign(device_ordinals, device_ordinals + n_devices);
    m->replica.assign(n_devices, nullptr);
    m->stream.assign(n_devices, nullptr);
    m->done.assign(n_devices, nullptr);
    m->d_tiles.assign(n_devices, nullptr);
    int rc = HRT_OK;
    for (uint32_t i = 0; i < n_devices && rc == HRT_OK; ++i) {
        rc = hrt_init(m->ordinal[i]);  // prepares the device on first use, makes it current
        if (rc == HRT_OK) rc = hrt_scene_create(desc, &m->replica[i]);
        if (rc == HRT_OK && hipStreamCreateWithFlags(&m->stream[i], hipStreamNonBlocking) != hipSuccess) rc = fail(HRT_ERR_DEVICE, "hrt_multi_create: hipStreamCreate failed");
        if (rc == HRT_OK && hipEventCreateWithFlags(&m->done[i], hipEventDisableTiming) != hipSuccess) rc = fail(HRT_ERR_DEVICE, "hrt_multi_create: hipEventCreate failed");
        if (rc == HRT_OK && i > 0 && m->ordinal[i] != m->ordinal[0]) {  // direct xGMI path for the gather where the topology offers it
            int can = 0;
            if (hipDeviceCanAccessPeer(&can, m->ordinal[0], m->ordinal[i]) == hipSuccess && can) {
                (void)hipSetDevice(m->ordinal[0]);
                const hipError_t e = hipDeviceEnablePeerAccess(m->ordinal[i], 0);
                if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) (void)hipGetLastError();  // staged copies still work
                else (void)hipGetLastError();
            }
        }
    }
    if (rc != HRT_OK) {
        const std::string keep = g_error;
        hrt_multi_destroy(m);
        g_error = keep;
        return rc;
    }
    (void)use_device(m->ordinal[0]);
    *out = m;
    return HRT_OK;
}

int hrt_multi_render(hrt_multi *m, const hrt_camera *cam, uint32_t w, uint32_t h, uint32_t spp, uint64_t seed, uint32_t flags,
                     float *out_rgb, hrt_stats *stats) {
    if (!m || !cam || !out_rgb) return fail(HRT_ERR_INVALID, "hrt_multi_render: NULL argument");
    if (!w || !h || !spp) return fail(HRT_ERR_INVALID, "render: w, h and spp must be positive");
    const auto t0 = std::chrono::steady_clock::now();
    const uint32_t n = m->n;
    const size_t per = (size_t)hrt_tiles_owned(w, h, 0, n) * 64u * 3u;  // floats per slot block, padded to slot 0's share
    const size_t frame_floats = (size_t)w * h * 3u;
    if (m->tiles_cap < per || m->frame_cap < frame_floats) {
        multi_free_buffers(m);
        HIP_TRY(hipSetDevice(m->ordinal[0]));
        HIP_TRY(hipMalloc((void **)&m->d_gathered, std::max<size_t>(per, 1) * n * sizeof(float)));
        HIP_TRY(hipMalloc((void **)&m->d_frame, frame_floats * sizeof(float)));
        for (uint32_t i = 1; i < n; ++i) {
            HIP_TRY(hipSetDevice(m->ordinal[i]));
            HIP_TRY(hipMalloc((void **)&m->d_tiles[i], std::max<size_t>(per, 1) * sizeof(float)));
        }
        m->tiles_cap = per;
        m->frame_cap = frame_floats;
    }
    // render: every slot at once; gather: each slot's block follows its kernel on the slot's stream
    for (uint32_t i = 0; i < n; ++i) {
        float *dst = m->d_gathered + (size_t)i * m->tiles_cap;
        float *tiles = i == 0 ? dst : m->d_tiles[i];
        const uint32_t owned = hrt_tiles_owned(w, h, i, n);
        int rc = hrt_render_tiles(m->replica[i], cam, w, h, spp, seed, flags, i, n, tiles, (void *)m->stream[i]);  // sets the slot's device
        if (rc != HRT_OK) return rc;
        if (i > 0 && owned)
            HIP_TRY(hipMemcpyPeerAsync(dst, m->ordinal[0], tiles, m->ordinal[i], (size_t)owned * 64u * 3u * sizeof(float), m->stream[i]));
        HIP_TRY(hipEventRecord(m->done[i], m->stream[i]));
    }
    int rc = use_device(m->ordinal[0]);
    if (rc != HRT_OK) return rc;
    for (uint32_t i = 1; i < n; ++i) HIP_TRY(hipStreamWaitEvent(m->stream[0], m->done[i], 0));
    rc = hrt_assemble_frame(m->d_gathered, (uint32_t)(m->tiles_cap / 192u), w, h, n, m->d_frame, (void *)m->stream[0]);
    if (rc != HRT_OK) return rc;
    HIP_TRY(hipMemcpyAsync(out_rgb, m->d_frame, frame_floats * sizeof(float), hipMemcpyDeviceToHost, m->stream[0]));
    HIP_TRY(hipStreamSynchronize(m->stream[0]));
    double kernel_ms = 0.0;
    for (uint32_t i = 0; i < n; ++i) {  // never hand back a frame a slot did not finish
        double ms = 0.0;
        rc = hrt_last_kernel_ms(m->replica[i], &ms);
        if (rc != HRT_OK) return rc;
        kernel_ms = std::max(kernel_ms, ms);
    }
    (void)use_device(m->ordinal[0]);
    if (stats) {
        std::memset(stats, 0, sizeof(*stats));
        stats->kernel_ms = kernel_ms;  // the slowest slot
        stats->total_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        stats->samples = (uint64_t)w * h * spp;
        stats->vgprs = (uint32_t)g_rt.attr.numRegs;
        stats->lds_bytes = m->replica[0]->last_lds;
        for (uint32_t i = 0; i < n; ++i) stats->waves_launched += m->replica[i]->last_waves;
    }
    return HRT_OK;
}

int hrt_render_multi(const hrt_scene_desc *desc, const hrt_camera *cam, uint32_t w, uint32_t h, uint32_t spp, uint64_t seed,
                     uint32_t flags, uint32_t n_devices, const int *device_ordinals, float *out_rgb, hrt_stats *stats) {
    hrt_multi *m = nullptr;
    int rc = hrt_multi_create(desc, n_devices, device_ordinals, &m);
    if (rc != HRT_OK) return rc;
    rc = hrt_multi_render(m, cam, w, h, spp, seed, flags, out_rgb, stats);
    const std::string keep = g_error;
    hrt_multi_destroy(m);
    g_error = keep;
    return rc;
}
