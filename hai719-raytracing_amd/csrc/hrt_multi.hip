// hrt_render_multi -- the whole frame on several GPUs from ONE process (include/hrt.h), the multi-GPU form of the
// reference's single caller ray_trace_from_camera() (main.cpp:200-263; its own fan-out is one std::thread per scanline,
// :232-238).  Included by hrt_api.hip inside its extern "C" block.
//
//   replicas   the scene is small (<= ~25 MB) and read-only: one hrt_scene per slot, on that slot's device
//   partition  8x8 tiles round-robin, slot r renders tiles r, r + N, ... (hrt_render_tiles), all slots at once, each on a
//              stream of its own; per-pixel RNG keys make the partition invisible in the pixels
//   gather     ONE step, no reduction (slots own disjoint pixels), behind every slot's kernel on the slot's own stream:
//                rccl   one communicator per slot (ncclCommInitAll at hrt_multi_create), one group of ncclGather calls with
//                       root 0 -- the collective the north star names; slot 0's block is in place.  Default whenever the
//                       ordinals are distinct (a communicator does not take a device twice); also with ONE slot, where
//                       the gather of one rank still goes through the communicator.
//                peer   every slot's dense tile buffer copied device to device (hipMemcpyPeerAsync: xGMI DMA between
//                       GPUs, a plain copy when a slot shares slot 0's device) into its block of slot 0's gather buffer.
//                       The form for repeated ordinals (several slots on one GPU: how a one-GPU box rehearses N slots),
//                       and selectable with HRT_MULTI_GATHER=peer.
//              The same bytes cross the same links either way; hrt_multi_gather() says which one a handle uses.
//   assemble   slot 0: tiles -> row-major frame (hrt_assemble_kernel), one D2H copy
//
// librccl.so is opened with dlopen when the first communicator is wanted: single-GPU users of libhrt.so never load it, and a
// process that already holds a copy (torch ships one) shares it.
// (<dlfcn.h> and <rccl/rccl.h> -- types only, nothing of it is linked -- are included at the top of hrt_api.hip.)
struct RcclApi {
    bool tried = false, ok = false;
    std::string why;
    decltype(&ncclCommInitAll) CommInitAll = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclGather) Gather = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
};
static RcclApi &rccl_api() {
    static RcclApi a;
    if (a.tried) return a;
    a.tried = true;
    void *lib = nullptr;
    for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"})
        if ((lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL))) break;
    if (!lib) { a.why = std::string("librccl.so could not be opened: ") + (dlerror() ? dlerror() : "?"); return a; }
#define HRT_RCCL_SYM(field, sym)                                                                    \
    a.field = reinterpret_cast<decltype(a.field)>(dlsym(lib, #sym));                                \
    if (!a.field) { a.why = "librccl.so lacks " #sym; return a; }
    HRT_RCCL_SYM(CommInitAll, ncclCommInitAll)
    HRT_RCCL_SYM(CommDestroy, ncclCommDestroy)
    HRT_RCCL_SYM(GroupStart, ncclGroupStart)
    HRT_RCCL_SYM(GroupEnd, ncclGroupEnd)
    HRT_RCCL_SYM(Gather, ncclGather)
    HRT_RCCL_SYM(GetErrorString, ncclGetErrorString)
#undef HRT_RCCL_SYM
    a.ok = true;
    return a;
}
#define RCCL_TRY(expr)                                                                                                  \
    do {                                                                                                                \
        const ncclResult_t r_ = (expr);                                                                                 \
        if (r_ != ncclSuccess) return fail(HRT_ERR_DEVICE, std::string(#expr) + ": " + rccl_api().GetErrorString(r_));  \
    } while (0)

struct hrt_multi {
    uint32_t n = 0;
    std::vector<int> ordinal;
    std::vector<hrt_scene *> replica;
    std::vector<hipStream_t> stream;
    std::vector<hipEvent_t> done;
    std::vector<bool> busy;            // `done` of the slot has been recorded by a render (and may still be pending)
    std::vector<float *> d_tiles;      // slots 1..n-1: this slot's tiles on its own device
    std::vector<ncclComm_t> comm;      // gather == rccl: one communicator per slot
    bool use_rccl = false;
    std::string note;                  // what creation fell back from, if anything (also left in hrt_last_error())
    size_t tiles_cap = 0;              // floats per slot buffer
    float *d_gathered = nullptr;       // slot 0's device: n blocks of tiles_cap floats
    float *d_frame = nullptr;
    size_t frame_cap = 0;
};

// Nothing of an earlier render may still be running when its buffers go away: slots on other devices run asynchronously.
static void multi_quiesce(hrt_multi *m) {
    for (uint32_t i = 0; i < m->n; ++i)
        if (i < m->busy.size() && m->busy[i] && m->done[i]) { (void)hipEventSynchronize(m->done[i]); m->busy[i] = false; }
}

static void multi_free_buffers(hrt_multi *m) {
    if (!m->n || !m->replica[0]) return;  // creation failed before any buffer existed
    multi_quiesce(m);
    for (uint32_t i = 0; i < m->n; ++i)
        if (i < m->d_tiles.size() && m->d_tiles[i] && use_device(m->ordinal[i]) == HRT_OK) { (void)hipFree(m->d_tiles[i]); m->d_tiles[i] = nullptr; }
    if (use_device(m->ordinal[0]) == HRT_OK) {
        if (m->d_gathered) (void)hipFree(m->d_gathered);
        if (m->d_frame) (void)hipFree(m->d_frame);
    }
    m->d_gathered = nullptr; m->d_frame = nullptr; m->tiles_cap = 0; m->frame_cap = 0;
}

void hrt_multi_destroy(hrt_multi *m) {
    if (!m) return;
    multi_free_buffers(m);
    for (uint32_t i = 0; i < m->n; ++i) {
        if (!m->replica[i]) continue;  // a slot hrt_multi_create never reached (e.g. a bad ordinal): nothing on it
        if (use_device(m->ordinal[i]) != HRT_OK) continue;
        if (i < m->comm.size() && m->comm[i]) (void)rccl_api().CommDestroy(m->comm[i]);
        if (i < m->stream.size() && m->stream[i]) (void)hipStreamDestroy(m->stream[i]);
        if (i < m->done.size() && m->done[i]) (void)hipEventDestroy(m->done[i]);
        hrt_scene_destroy(m->replica[i]);
    }
    if (m->n && g_rt.ready && m->replica[0]) (void)use_device(m->ordinal[0]);
    (void)hipGetLastError();  // nothing above may leave a sticky error for the next launch check
    delete m;
}

const char *hrt_multi_gather(const hrt_multi *m) { return !m ? "" : (m->use_rccl ? "rccl" : "peer"); }

int hrt_multi_create(const hrt_scene_desc *desc, uint32_t n_devices, const int *device_ordinals, hrt_multi **out) {
    if (!desc || !out || !n_devices || !device_ordinals) return fail(HRT_ERR_INVALID, "hrt_multi_create: bad argument");
    if (n_devices > 64u) return fail(HRT_ERR_INVALID, "hrt_multi_create: more than 64 slots");
    bool distinct = true;
    for (uint32_t i = 0; i < n_devices; ++i)
        for (uint32_t j = 0; j < i; ++j) distinct = distinct && device_ordinals[i] != device_ordinals[j];
    const char *env = std::getenv("HRT_MULTI_GATHER");
    const std::string want = env ? env : "";
    if (!want.empty() && want != "rccl" && want != "peer") return fail(HRT_ERR_INVALID, "HRT_MULTI_GATHER must be rccl or peer");
    if (want == "rccl" && !distinct) return fail(HRT_ERR_INVALID, "hrt_multi_create: HRT_MULTI_GATHER=rccl needs distinct devices (a communicator does not take a device twice)");
    hrt_multi *m = new hrt_multi();
    m->n = n_devices;
    m->ordinal.assign(device_ordinals, device_ordinals + n_devices);
    m->replica.assign(n_devices, nullptr);
    m->stream.assign(n_devices, nullptr);
    m->done.assign(n_devices, nullptr);
    m->busy.assign(n_devices, false);
    m->d_tiles.assign(n_devices, nullptr);
    m->comm.assign(n_devices, nullptr);
    int rc = HRT_OK;
    for (uint32_t i = 0; i < n_devices && rc == HRT_OK; ++i) {
        rc = hrt_init(m->ordinal[i]);  // prepares the device on first use, makes it current
        if (rc == HRT_OK) rc = hrt_scene_create(desc, &m->replica[i]);
        if (rc == HRT_OK && hipStreamCreateWithFlags(&m->stream[i], hipStreamNonBlocking) != hipSuccess) rc = fail(HRT_ERR_DEVICE, "hrt_multi_create: hipStreamCreate failed");
        if (rc == HRT_OK && hipEventCreateWithFlags(&m->done[i], hipEventDisableTiming) != hipSuccess) rc = fail(HRT_ERR_DEVICE, "hrt_multi_create: hipEventCreate failed");
    }
    if (rc == HRT_OK && want != "peer" && distinct) {  // the collective
        RcclApi &api = rccl_api();
        if (!api.ok) {
            if (want == "rccl") rc = fail(HRT_ERR_DEVICE, "hrt_multi_create: " + api.why);
            else m->note = "gather falls back to peer copies: " + api.why;
        } else {
            const ncclResult_t r = api.CommInitAll(m->comm.data(), (int)n_devices, m->ordinal.data());
            if (r != ncclSuccess) {
                for (auto &c : m->comm) c = nullptr;
                (void)hipGetLastError();
                if (want == "rccl") rc = fail(HRT_ERR_DEVICE, std::string("hrt_multi_create: ncclCommInitAll: ") + api.GetErrorString(r));
                else m->note = std::string("gather falls back to peer copies: ncclCommInitAll: ") + api.GetErrorString(r);
            } else {
                m->use_rccl = true;
            }
        }
    }
    if (rc == HRT_OK && !m->use_rccl)  // peer copies: the direct xGMI path where the topology offers it, in both directions
        for (uint32_t i = 1; i < n_devices && rc == HRT_OK; ++i) {
            if (m->ordinal[i] == m->ordinal[0]) continue;
            for (int dir = 0; dir < 2 && rc == HRT_OK; ++dir) {
                const int self = dir ? m->ordinal[i] : m->ordinal[0], peer = dir ? m->ordinal[0] : m->ordinal[i];
                int can = 0;
                if (hipDeviceCanAccessPeer(&can, self, peer) != hipSuccess || !can) {
                    (void)hipGetLastError();
                    m->note += "no peer access " + std::to_string(self) + " -> " + std::to_string(peer) + " (copies are staged); ";
                    continue;
                }
                rc = use_device(self);
                if (rc != HRT_OK) break;
                const hipError_t e = hipDeviceEnablePeerAccess(peer, 0);
                (void)hipGetLastError();
                if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled)  // hipMemcpyPeerAsync still works, staged through the host
                    m->note += "hipDeviceEnablePeerAccess(" + std::to_string(self) + " -> " + std::to_string(peer) + "): " + hipGetErrorString(e) + " (copies are staged); ";
            }
        }
    if (rc == HRT_OK) rc = use_device(m->ordinal[0]);
    if (rc != HRT_OK) {
        const std::string keep = g_error;
        hrt_multi_destroy(m);
        g_error = keep;
        return rc;
    }
    g_error = m->note;  // HRT_OK, but what was fallen back from is readable through hrt_last_error()
    *out = m;
    return HRT_OK;
}

int hrt_multi_render(hrt_multi *m, const hrt_camera *cam, uint32_t w, uint32_t h, uint32_t spp, uint64_t seed, uint32_t flags,
                     float *out_rgb, hrt_stats *stats) {
    if (!m || !cam || !out_rgb) return fail(HRT_ERR_INVALID, "hrt_multi_render: NULL argument");
    if (!w || !h || !spp) return fail(HRT_ERR_INVALID, "render: w, h and spp must be positive");
    const auto t0 = std::chrono::steady_clock::now();
    const uint32_t n = m->n;
    const size_t per = std::max<size_t>((size_t)hrt_tiles_owned(w, h, 0, n) * 64u * 3u, 1);  // floats per slot block: slot 0's share is the largest
    const size_t frame_floats = (size_t)w * h * 3u;
    int rc;
    if (m->tiles_cap < per || m->frame_cap < frame_floats) {
        multi_free_buffers(m);  // waits for every slot's previous launch and copy first
        if ((rc = use_device(m->ordinal[0])) != HRT_OK) return rc;
        HIP_TRY(hipMalloc((void **)&m->d_gathered, per * n * sizeof(float)));
        HIP_TRY(hipMalloc((void **)&m->d_frame, frame_floats * sizeof(float)));
        for (uint32_t i = 1; i < n; ++i) {
            if ((rc = use_device(m->ordinal[i])) != HRT_OK) return rc;
            HIP_TRY(hipMalloc((void **)&m->d_tiles[i], per * sizeof(float)));
        }
        m->tiles_cap = per;
        m->frame_cap = frame_floats;
        if ((rc = use_device(m->ordinal[0])) != HRT_OK) return rc;
    }
    // render: every slot at once, blocks of `per` floats in the gather buffer
    for (uint32_t i = 0; i < n; ++i) {
        float *tiles = i == 0 ? m->d_gathered : m->d_tiles[i];
        rc = hrt_render_tiles(m->replica[i], cam, w, h, spp, seed, flags, i, n, tiles, (void *)m->stream[i]);  // switches to the slot's device
        if (rc != HRT_OK) return rc;
    }
    // gather: behind each slot's kernel, on the slot's stream
    if (m->use_rccl) {
        RcclApi &api = rccl_api();
        RCCL_TRY(api.GroupStart());
        for (uint32_t i = 0; i < n; ++i) {
            const float *src = i == 0 ? m->d_gathered : m->d_tiles[i];  // slot 0 in place: sendbuff == recvbuff + 0 * per
            const ncclResult_t r = api.Gather(src, i == 0 ? m->d_gathered : nullptr, per, ncclFloat, 0, m->comm[i], m->stream[i]);
            if (r != ncclSuccess) { (void)api.GroupEnd(); return fail(HRT_ERR_DEVICE, std::string("ncclGather: ") + api.GetErrorString(r)); }
        }
        RCCL_TRY(api.GroupEnd());
    } else {
        for (uint32_t i = 1; i < n; ++i) {
            const uint32_t owned = hrt_tiles_owned(w, h, i, n);
            if (!owned) continue;
            if ((rc = use_device(m->ordinal[i])) != HRT_OK) return rc;
            HIP_TRY(hipMemcpyPeerAsync(m->d_gathered + (size_t)i * per, m->ordinal[0], m->d_tiles[i], m->ordinal[i], (size_t)owned * 64u * 3u * sizeof(float), m->stream[i]));
        }
    }
    for (uint32_t i = 0; i < n; ++i) {
        if ((rc = use_device(m->ordinal[i])) != HRT_OK) return rc;
        HIP_TRY(hipEventRecord(m->done[i], m->stream[i]));
        m->busy[i] = true;
    }
    if ((rc = use_device(m->ordinal[0])) != HRT_OK) return rc;
    for (uint32_t i = 1; i < n; ++i) HIP_TRY(hipStreamWaitEvent(m->stream[0], m->done[i], 0));
    rc = hrt_assemble_frame(m->d_gathered, (uint32_t)(per / 192u), w, h, n, m->d_frame, (void *)m->stream[0]);
    if (rc != HRT_OK) return rc;
    HIP_TRY(hipMemcpyAsync(out_rgb, m->d_frame, frame_floats * sizeof(float), hipMemcpyDeviceToHost, m->stream[0]));
    HIP_TRY(hipStreamSynchronize(m->stream[0]));
    for (uint32_t i = 0; i < n; ++i) m->busy[i] = false;  // stream 0 waited for every slot's `done`
    double kernel_ms = 0.0;
    for (uint32_t i = 0; i < n; ++i) {  // never hand back a frame a slot did not finish
        double ms = 0.0;
        rc = hrt_last_kernel_ms(m->replica[i], &ms);
        if (rc != HRT_OK) return rc;
        kernel_ms = std::max(kernel_ms, ms);
    }
    if ((rc = use_device(m->ordinal[0])) != HRT_OK) return rc;
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
