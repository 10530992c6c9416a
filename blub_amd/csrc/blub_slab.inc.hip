// z-slab domain decomposition of the fluid step -- protocol + transports.  Included by blub_fluid.hip.
//
// One protocol, two transports:
//   loopback : all slabs live in this process on one GPU and share one stream (validation of the protocol on a 1-GPU box);
//              halo / particle transfers are device-to-device copies, the all-reduce is a tiny kernel
//   RCCL     : one slab per process / GPU; halo planes and particles travel as grouped ncclSend/ncclRecv between
//              z-neighbours (point-to-point over xGMI), the PCG partials as ncclSend/ncclRecv to every other rank inside
//              the same group (an all-gather spelled as p2p so that it fuses with the halo into one operation)
// Per step: 3 particle exchanges, 5 three-volume halo exchanges, 2 x (descriptor + r + s + p) halos and per PCG iteration TWO
// grouped point-to-point operations: the s.As partials after the direction kernel, and the r plane + {(M^-1 r).r, max|r|}
// partials after the update kernel.  The iteration count follows the previous solve (+ re-checks), see slab_solve.
#include <rccl/rccl.h>

#include <functional>

struct blub_slab_group {
    std::vector<blub_fluid*> slabs;   // local slabs, ascending z
    int nranks = 1, first = 0;        // total number of slabs / global index of slabs[0]
    bool rccl = false;
    ncclComm_t comm = nullptr;
    hipStream_t stream = nullptr;
    int device = 0;
    uint32_t capacity = 0;            // particle capacity of every slab and of the transfer buffers
    int mem_mode = 0;                 // BLUB_SLAB_MEMORY_*: what the exportable regions were allocated with
    bool full_volumes = false;        // every slab holds the whole grid (BLUB_SLAB_FULL_VOLUMES): the precondition of blub_slab_group_recut
    float* layer_hist = nullptr;      // device, nranks x layers: FLUID bricks per brick layer, one segment per rank (blub_slab_group_rebalance)
    float* layer_hist_host = nullptr; // pinned
    std::vector<int> cuts;            // nranks + 1 cut planes (multiples of the brick depth): slab r owns [cuts[r], cuts[r + 1]); uniform unless the caller passed its own
    std::vector<int> vol_z0_of; std::vector<size_t> vol_first_of;   // per rank: first plane its volumes hold / that plane's first cell (blub_fluid::vol_z0, vol_first)
    struct Extra {
        uint32_t *leave_idx = nullptr, *hole_idx = nullptr, *fill_idx = nullptr;   // in-place migration (blub_slab.hip.h: k_slab_migrate_*)
        float4 *up[4] = {nullptr, nullptr, nullptr, nullptr}, *dn[4] = {nullptr, nullptr, nullptr, nullptr};   // send buffers: pos, vx, vy, vz
        float4 *up_msg = nullptr, *dn_msg = nullptr;   // allocations behind up[0] / dn[0]: one float4 of header in front of the position payload
        float4 *rb_msg = nullptr, *ra_msg = nullptr, *rb[4] = {nullptr, nullptr, nullptr, nullptr}, *ra[4] = {nullptr, nullptr, nullptr, nullptr};   // staging of what arrives from below / above
        uint32_t* append_done = nullptr;         // device: block counter of k_slab_append
        blubk::SlabCounts* counts = nullptr;     // device
        uint32_t* recv_counts = nullptr;         // device: {from below, from above}
        float* gat_dir = nullptr;                // device: nranks x SLAB_NP partials of s.As (own segment written by the direction kernel)
        float2* gat_upd = nullptr;               // device: nranks x SLAB_NP partials {(M^-1 r).r, max|r|} (own segment written by init / update)
        float4* gat4[2] = {nullptr, nullptr};    // device: nranks x SLAB_NP partials {gamma, delta, max|r|} of the single-reduction schedule, by iteration parity
        float* gat_cnt = nullptr;                // device: nranks fluid-brick counts (own entry written at the start of a step)
    };
    std::vector<Extra> ex;
    blubk::SlabCounts* counts_host = nullptr;    // pinned, one per local slab
    uint32_t* recv_host = nullptr;               // pinned, two per local slab
    blubk::PcgCtrl* ctrl_host = nullptr;         // pinned: control block of slab 0's solves [velocity, density], read only after a stream sync
    bool ctrl_host_valid[2] = {false, false};
    uint64_t comm_ops = 0;                       // grouped transport operations issued so far (diagnostics)
    float* cnt_host = nullptr;                   // pinned: [0, nranks) gathered fluid-brick counts, [nranks, nranks + nlocal) staging of the own ones
    int np_cur = blubk::SLAB_NP_DEFAULT;         // PCG grid of this step's slab solves
    blubk::SlabCopyList copies{};                // loopback transport: plane copies collected for one batched launch
    // host-synchronisation-free particle exchange (slab_exchange_particles_async): per local slab and exchange kind what travelled last time
    struct Hist { bool valid = false, pending = false; uint32_t seq = 0, n_up = 0, n_down = 0, from_below = 0, from_above = 0; };
    std::vector<Hist> hist;                      // [local slab][XFER_KINDS]
    blubk::SlabXferRecord* rec_host = nullptr;   // pinned: [local slab][XFER_KINDS][XFER_RING] records written by k_slab_append
    blubk::SlabXferRecord* rec_dev = nullptr;    // the same ring as the device sees it
    uint32_t xfer_seq = 0;                       // exchanges issued so far
    bool async_exchange = true;                  // blub_slab_group_set_async_exchange
    float* cntrec_host = nullptr; float* cntrec_dev = nullptr;       // pinned: [2][nranks] gathered fluid-brick counts, double buffered by step parity
    uint32_t* cntseq_host = nullptr; uint32_t* cntseq_dev = nullptr; // pinned: [2] tags
    uint32_t cnt_seq = 0; bool cnt_pending = false;
    uint64_t host_syncs = 0, done_polls = 0;     // stream synchronisations issued by blub_slab_group_step so far: particle exchanges / looks at a solve's `done` (diagnostics)
    int gather_mode = 0;                         // RCCL only: 0 = partials as p2p inside the halo's group, 1 = ncclAllGather (calibrated at creation)
    char transport[192] = "loopback";
    // ---- DIRECT transport (round 4): every slab writes what its neighbours need straight into THEIR memory (the same process in a local
    // group, hipIpc mappings between processes) and raises a flag word; consumers wait for flags -- inside the PCG iteration kernel, with a
    // one-block wait kernel elsewhere.  No host-issued transport operation, no stream synchronisation, no message sizes.
    bool direct = false;
    struct Region { char* base = nullptr; size_t bytes = 0; };
    std::vector<std::vector<Region>> regions;       // [local slab][k]: exportable allocations, the same list on every rank
    std::vector<std::vector<char*>> peer_base;      // [rank][k]: region k of that rank's slab as mapped in THIS process (own slabs: the pointer itself)
    std::vector<void*> ipc_opened;                  // mappings to close
    struct Arena { char* base = nullptr; size_t bytes = 0, used = 0; };
    std::vector<Arena> arena;                       // [local slab]: flags, gather arrays, particle staging (one allocation: one hipIpc handle)
    std::vector<uint32_t*> flags, blocks_done, dir_error;   // [local slab]: flag words (one per source rank), finished-workgroup counter, time-out marker
    uint32_t flag_seq = 0;                          // exchanges issued so far (the same number on every rank)
    blubk::SlabFlagList push_flags{};               // flags the pending batched push raises
    // ---- checkpoints / in-place recovery (round 5): two generations of the restartable state per local slab, taken every `ck_interval` steps
    uint32_t ck_interval = 0;
    std::vector<blubk::SlabCheckpoint> ck[2];       // [generation][local slab]
    std::vector<void*> ck_allocs;
};

namespace blub {

#define NCCL_TRY(expr)                                                                                          \
    do {                                                                                                        \
        ncclResult_t _r = (expr);                                                                               \
        if (_r != ncclSuccess) { char _b[256]; snprintf(_b, sizeof _b, "%s failed: %s", #expr, ncclGetErrorString(_r)); return set_error(BLUB_ERR_COMM, _b); } \
    } while (0)

// exchange sequence numbers never take the value 0: flag / acknowledgement words start at 0 and a partial's tag 0 means "never written" (k_pcg1_finalize);
// ordering is by signed difference, so the wrap after 2^32 exchanges (~14 h of stepping at 1200 steps/s) is harmless otherwise (round-4 ADVICE)
static inline uint32_t seq_after(uint32_t s) { return s + 1u ? s + 1u : 1u; }

enum { XFER_GHOST_FULL = 0, XFER_GHOST_POS = 1, XFER_MIGRATE = 2, XFER_MIGRATE_B = 3, XFER_KINDS = 4, XFER_RING = 4 };   // (MIGRATE_B: the second migration of a step; same protocol, its own history)

// loopback transport: device-to-device plane copies are collected and issued as ONE kernel per exchange
// launch width of a batch: the largest copy decides (items = 16-byte, 4-byte or 1-byte units, see SlabCopy)
static unsigned slab_copy_width(const blub_slab_group* G) {
    uint32_t mx = 0;
    for (int k = 0; k < G->copies.n; ++k) { const auto& c = G->copies.c[k]; mx = std::max(mx, c.unit == 0u ? c.bytes / 16u : (c.unit == 1u ? c.bytes / 4u : c.bytes)); }
    return std::max(1u, std::min(64u, (mx + 255u) / 256u));
}
static int slab_copy_flush(blub_slab_group* G) {
    if (G->copies.n == 0 && !(G->direct && G->push_flags.n)) return BLUB_OK;
    const unsigned bx = slab_copy_width(G);
    if (G->direct) {      // write-through stores, then the flags of this exchange
        G->push_flags.seq = G->flag_seq; G->push_flags.blocks_done = G->blocks_done[0];
        hipLaunchKernelGGL(blubk::k_slab_push_planes, dim3(bx, (unsigned)std::max(1, G->copies.n)), dim3(256), 0, G->stream, G->copies, G->push_flags);
        G->push_flags.n = 0; G->push_flags.n_ack = 0;
    } else
        hipLaunchKernelGGL(blubk::k_slab_copy_planes, dim3(bx, (unsigned)G->copies.n), dim3(256), 0, G->stream, G->copies);
    G->copies.n = 0;
    return BLUB_OK;
}
static int slab_copy(blub_slab_group* G, void* dst, const void* src, size_t bytes) {
    if (bytes == 0) return BLUB_OK;
    if (bytes > 0xFFFFFFF0u) {
        // (no plane or message of a supported grid is this large; the direct transport must not leave its one push kernel -- a plain copy into a peer's
        //  memory would skip the acknowledgement handshake and a flush here would raise this exchange's flags early: round-4 ADVICE)
        if (G->direct) return set_error(BLUB_ERR_UNSUPPORTED, "direct transport: a single transfer of 4 GiB or more");
        int rc = slab_copy_flush(G);
        if (rc != BLUB_OK) return rc;
        HIP_TRY(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, G->stream));
        return BLUB_OK;
    }
    if (G->copies.n == blubk::SLAB_COPY_MAX) {
        if (G->direct) {      // (a partial flush must not raise this exchange's flags: plain batch, flags with the last one)
            const blubk::SlabFlagList keep = G->push_flags; G->push_flags.n = 0;
            blubk::SlabFlagList none = keep; none.n = 0; none.blocks_done = G->blocks_done[0]; none.seq = G->flag_seq;      // (no flags yet; the acknowledgement handshake of `keep` runs in front of these stores too)
            hipLaunchKernelGGL(blubk::k_slab_push_planes, dim3(slab_copy_width(G), (unsigned)G->copies.n), dim3(256), 0, G->stream, G->copies, none);
            G->copies.n = 0; G->push_flags = keep;
        } else { int rc = slab_copy_flush(G); if (rc != BLUB_OK) return rc; }
    }
    blubk::SlabCopy& c = G->copies.c[G->copies.n];
    const uintptr_t al = (uintptr_t)dst | (uintptr_t)src | (uintptr_t)bytes;
    c.src = src; c.dst = dst; c.bytes = (uint32_t)bytes; c.unit = (al & 15) == 0 ? 0u : ((al & 3) == 0 ? 1u : 2u);
    G->copies.n += 1;
    return BLUB_OK;
}

static bool has_up(const blub_slab_group* G, int i) { return G->first + i + 1 < G->nranks; }
static bool has_down(const blub_slab_group* G, int i) { return G->first + i > 0; }
static bool up_local(const blub_slab_group* G, int i) { return i + 1 < (int)G->slabs.size(); }
static bool down_local(const blub_slab_group* G, int i) { return i > 0; }

// ---- DIRECT transport: address translation and flags ----------------------------------------------------------------------------------
// `mine` points into one of local slab i's exportable allocations; the same offset inside rank r's allocation of the same kind
template <class T>
static T* peer_ptr(const blub_slab_group* G, int i, int r, T* mine) {
    const char* m = reinterpret_cast<const char*>(mine);
    const auto& regs = G->regions[i];
    for (size_t k = 0; k < regs.size(); ++k)
        if (m >= regs[k].base && m < regs[k].base + regs[k].bytes) return reinterpret_cast<T*>(G->peer_base[r][k] + (m - regs[k].base));
    return nullptr;
}
// ... and for a grid VOLUME, whose pointer is that of plane 0 while the allocation starts at the owner's first held plane (blub_fluid::vol_first):
// the peer's plane-0 pointer of the same volume
template <class T>
static T* peer_vol(const blub_slab_group* G, int i, int r, T* mine, size_t elem = sizeof(T)) {
    char* real = peer_ptr(G, i, r, reinterpret_cast<char*>(mine) + G->slabs[i]->vol_first * elem);
    return real ? reinterpret_cast<T*>(real - G->vol_first_of[(size_t)r] * elem) : nullptr;
}
static bool rank_local(const blub_slab_group* G, int r) { return r >= G->first && r < G->first + (int)G->slabs.size(); }
// the flag word of rank `dst` for messages from rank `src` (dst's flags live in dst's arena; translated through local slab i = the sender)
static uint32_t* flag_of(const blub_slab_group* G, int i, int dst) { return peer_ptr(G, i, dst, G->flags[i]) + (G->first + i); }
static int slab_push_flush(blub_slab_group* G);
// a local slab's stream waits for the flags of the exchange just issued (not needed when every source is a local slab: one stream orders them)
static int slab_wait(blub_slab_group* G, int i, uint32_t mask, const blubk::PcgCtrl* skip_if_done = nullptr) {
    uint32_t remote = 0;
    for (int r = 0; r < G->nranks; ++r) if (((mask >> r) & 1u) && !rank_local(G, r)) remote |= 1u << r;
    if (!remote) return BLUB_OK;
    hipLaunchKernelGGL(blubk::k_slab_wait, dim3(1), dim3(64), 0, G->stream, (const uint32_t*)G->flags[i], remote, G->flag_seq, G->dir_error[i], skip_if_done);
    return BLUB_OK;
}
static uint32_t neighbour_mask(const blub_slab_group* G, int i) { return (has_up(G, i) ? 1u << (G->first + i + 1) : 0u) | (has_down(G, i) ? 1u << (G->first + i - 1) : 0u); }
static uint32_t others_mask(const blub_slab_group* G, int i) { return ((G->nranks >= 32 ? 0xFFFFFFFFu : (1u << G->nranks) - 1u)) & ~(1u << (G->first + i)); }
// acknowledgement words live behind the flag words of a slab's flag array: [32 + source rank]
constexpr int SLAB_ACK_OFFSET = 32;
static void flag_list_add(const blub_slab_group* G, blubk::SlabFlagList& F, int i, int dst) {
    uint32_t* f = flag_of(G, i, dst);
    for (int k = 0; k < F.n; ++k) if (F.f[k] == f) return;
    if (F.n < blubk::SLAB_FLAG_MAX) F.f[F.n++] = f;
    if (!rank_local(G, dst) && F.n_ack < 8) {      // another process: handshake before writing into its memory (blub_slab.hip.h: slab_ack_handshake)
        F.ack_out[F.n_ack] = f + SLAB_ACK_OFFSET; F.ack_src[F.n_ack] = dst; F.n_ack += 1;
        F.ack_in = G->flags[i] + SLAB_ACK_OFFSET; F.error = G->dir_error[i];
    }
}
static void push_flag_to(blub_slab_group* G, int i, int dst) { flag_list_add(G, G->push_flags, i, dst); }

// One z-plane of a volume from each z-neighbour: plane z1-1 goes up, plane z0 goes down; the receiver stores it at the
// same global z (ghost planes z0-1 and z1).
static int slab_halo(blub_slab_group* G, const std::vector<std::function<void*(blub_fluid*)>>& fields, size_t elem, bool own_group = true) {
    const blub_fluid* h0 = G->slabs[0];
    const size_t pb = (size_t)h0->g.nx * h0->g.ny * elem;
    if (own_group) G->comm_ops += 1;
    if (G->direct) {
        // every slab PUSHES its two boundary planes into its z-neighbours' ghost planes, the last workgroup of the batched launch raises the
        // neighbours' flags, the consumers' stream waits for its own (slab_wait)
        if (own_group) G->flag_seq = seq_after(G->flag_seq);
        for (int i = 0; i < (int)G->slabs.size(); ++i) {
            blub_fluid* h = G->slabs[i];
            for (auto& f : fields) {
                char* base = (char*)f(h);
                if (has_up(G, i)) { int rc = slab_copy(G, peer_vol(G, i, G->first + i + 1, base, elem) + (size_t)(h->slab_z1 - 1) * pb, base + (size_t)(h->slab_z1 - 1) * pb, pb); if (rc != BLUB_OK) return rc; }
                if (has_down(G, i)) { int rc = slab_copy(G, peer_vol(G, i, G->first + i - 1, base, elem) + (size_t)h->slab_z0 * pb, base + (size_t)h->slab_z0 * pb, pb); if (rc != BLUB_OK) return rc; }
            }
            if (has_up(G, i)) push_flag_to(G, i, G->first + i + 1);
            if (has_down(G, i)) push_flag_to(G, i, G->first + i - 1);
        }
        if (!own_group) return BLUB_OK;
        { int rc = slab_copy_flush(G); if (rc != BLUB_OK) return rc; }
        for (int i = 0; i < (int)G->slabs.size(); ++i) { int rc = slab_wait(G, i, neighbour_mask(G, i)); if (rc != BLUB_OK) return rc; }
        return BLUB_OK;
    }
    if (G->rccl && own_group) NCCL_TRY(ncclGroupStart());
    for (int i = 0; i < (int)G->slabs.size(); ++i) {
        blub_fluid* h = G->slabs[i];
        for (auto& f : fields) {
            char* base = (char*)f(h);
            if (has_up(G, i)) {
                if (up_local(G, i)) {
                    char* nb = (char*)f(G->slabs[i + 1]);
                    { int rc = slab_copy(G, nb + (size_t)(h->slab_z1 - 1) * pb, base + (size_t)(h->slab_z1 - 1) * pb, pb); if (rc != BLUB_OK) return rc; }
                    { int rc = slab_copy(G, base + (size_t)h->slab_z1 * pb, nb + (size_t)h->slab_z1 * pb, pb); if (rc != BLUB_OK) return rc; }
                } else {
                    NCCL_TRY(ncclSend(base + (size_t)(h->slab_z1 - 1) * pb, pb, ncclChar, G->first + i + 1, G->comm, G->stream));
                    NCCL_TRY(ncclRecv(base + (size_t)h->slab_z1 * pb, pb, ncclChar, G->first + i + 1, G->comm, G->stream));
                }
            }
            if (has_down(G, i) && !down_local(G, i)) {
                NCCL_TRY(ncclSend(base + (size_t)h->slab_z0 * pb, pb, ncclChar, G->first + i - 1, G->comm, G->stream));
                NCCL_TRY(ncclRecv(base + (size_t)(h->slab_z0 - 1) * pb, pb, ncclChar, G->first + i - 1, G->comm, G->stream));
            }
        }
    }
    if (G->rccl && own_group) NCCL_TRY(ncclGroupEnd());
    return own_group ? slab_copy_flush(G) : BLUB_OK;   // (inside slab_fused the caller flushes)
}
static int slab_halo_velocity(blub_slab_group* G) {
    return slab_halo(G, {[](blub_fluid* h) { return (void*)h->vel[0]; }, [](blub_fluid* h) { return (void*)h->vel[1]; }, [](blub_fluid* h) { return (void*)h->vel[2]; }}, 4);
}

// Segment `rank` of a gather array (seg_floats floats per slab) to every other slab.
static int slab_gather(blub_slab_group* G, const std::function<float*(int)>& array_of, int seg_floats, bool own_group = true) {
    if (G->nranks == 1) return BLUB_OK;
    if (own_group) G->comm_ops += 1;
    if (G->direct) {      // every slab pushes its own segment into every other slab's array
        if (own_group) G->flag_seq = seq_after(G->flag_seq);
        for (int i = 0; i < (int)G->slabs.size(); ++i) {
            float* mine = array_of(i) + (size_t)(G->first + i) * seg_floats;
            for (int r = 0; r < G->nranks; ++r) {
                if (r == G->first + i) continue;
                int rc = slab_copy(G, peer_ptr(G, i, r, mine), mine, (size_t)seg_floats * sizeof(float));
                if (rc != BLUB_OK) return rc;
                push_flag_to(G, i, r);
            }
        }
        if (!own_group) return BLUB_OK;
        { int rc = slab_copy_flush(G); if (rc != BLUB_OK) return rc; }
        for (int i = 0; i < (int)G->slabs.size(); ++i) { int rc = slab_wait(G, i, others_mask(G, i)); if (rc != BLUB_OK) return rc; }
        return BLUB_OK;
    }
    if (!G->rccl) {   // loopback: segment s of slab s's array into every other slab's array, batched with the plane copies of the same exchange
        const int S = (int)G->slabs.size();
        for (int sidx = 0; sidx < S; ++sidx)
            for (int d = 0; d < S; ++d) {
                if (d == sidx) continue;
                int rc = slab_copy(G, array_of(d) + (size_t)sidx * seg_floats, array_of(sidx) + (size_t)sidx * seg_floats, (size_t)seg_floats * sizeof(float));
                if (rc != BLUB_OK) return rc;
            }
        return own_group ? slab_copy_flush(G) : BLUB_OK;
    }
    float* arr = array_of(0);
    if (G->gather_mode == 1) {   // (never inside a p2p group: see slab_fused)
        NCCL_TRY(ncclAllGather(arr + (size_t)G->first * seg_floats, arr, (size_t)seg_floats, ncclFloat, G->comm, G->stream));
        return BLUB_OK;
    }
    if (own_group) NCCL_TRY(ncclGroupStart());
    for (int q = 0; q < G->nranks; ++q) {
        if (q == G->first) continue;
        NCCL_TRY(ncclSend(arr + (size_t)G->first * seg_floats, (size_t)seg_floats, ncclFloat, q, G->comm, G->stream));
        NCCL_TRY(ncclRecv(arr + (size_t)q * seg_floats, (size_t)seg_floats, ncclFloat, q, G->comm, G->stream));
    }
    if (own_group) NCCL_TRY(ncclGroupEnd());
    return BLUB_OK;
}
// Halo planes + a partial gather as ONE grouped operation (p2p mode) or as a p2p group followed by an all-gather.
static int slab_fused(blub_slab_group* G, const std::function<int()>& halos, const std::function<int(bool)>& gather) {
    int rc;
    G->comm_ops += 1;
    if (G->direct) {      // planes + partial segments in ONE batched push, one flag round
        G->flag_seq = seq_after(G->flag_seq);
        if ((rc = halos()) != BLUB_OK) return rc;
        if ((rc = gather(false)) != BLUB_OK) return rc;
        if ((rc = slab_copy_flush(G)) != BLUB_OK) return rc;
        for (int i = 0; i < (int)G->slabs.size(); ++i) if ((rc = slab_wait(G, i, others_mask(G, i))) != BLUB_OK) return rc;
        return BLUB_OK;
    }
    const bool split = G->rccl && G->gather_mode == 1;
    if (G->rccl) NCCL_TRY(ncclGroupStart());
    if ((rc = halos()) != BLUB_OK) return rc;
    if (!split && (rc = gather(false)) != BLUB_OK) return rc;
    if ((rc = slab_copy_flush(G)) != BLUB_OK) return rc;      // loopback: planes + partial segments in one launch
    if (G->rccl) NCCL_TRY(ncclGroupEnd());
    if (split) { G->comm_ops += 1; if ((rc = gather(false)) != BLUB_OK) return rc; }
    return BLUB_OK;
}

// Ghost copies / migration of particles between z-neighbours.
// device copy of the (exact) host-side particle counts of a slab: {own, ghost, overflow flag}
static int slab_upload_counts(blub_slab_group* G, int i) {
    blub_fluid* h = G->slabs[i];
    const uint32_t v[2] = {h->num_particles, h->num_ghost};
    // (pageable source: the runtime stages it before returning, so the stack array may go out of scope)
    HIP_TRY(hipMemcpyAsync(h->n_dev, v, sizeof v, hipMemcpyHostToDevice, G->stream));
    return BLUB_OK;
}

// Synchronous variant: the counts travel first and the host waits for them (one stream synchronisation per exchange).  Used for the first
// step after the particles were set from outside (no history to size messages from) and when the asynchronous exchange is switched off.
static int slab_exchange_particles(blub_slab_group* G, int kind) {
    const int S = (int)G->slabs.size();
    const int mode = kind == XFER_MIGRATE_B ? XFER_MIGRATE : kind;
    const bool with_rows = mode != XFER_GHOST_POS;
    for (int i = 0; i < S; ++i) {
        blub_fluid* h = G->slabs[i];
        auto& e = G->ex[i];
        HIP_TRY(hipMemsetAsync(e.counts, 0, sizeof(blubk::SlabCounts), G->stream));
        const uint32_t n = h->num_particles;
        if (!n) continue;
        if (mode == XFER_MIGRATE) {
            hipLaunchKernelGGL(blubk::k_slab_migrate_mark, dim3(particle_blocks(n)), dim3(256), 0, G->stream, n, (const float4*)h->pos, (const float4*)h->pvel[0], (const float4*)h->pvel[1],
                               (const float4*)h->pvel[2], (float)h->slab_z0, (float)h->slab_z1, G->capacity, e.counts, e.up[0], e.up[1], e.up[2], e.up[3], e.dn[0], e.dn[1], e.dn[2], e.dn[3], e.leave_idx,
                               (const uint32_t*)nullptr, G->capacity, G->capacity, h->pos);
            // (grids cover the worst case -- every particle leaves --; blocks beyond the device-side counts exit at once)
            hipLaunchKernelGGL(blubk::k_slab_migrate_match, dim3(particle_blocks(n)), dim3(256), 0, G->stream, n, (const float4*)h->pos, (float)h->slab_z0, (float)h->slab_z1, e.counts,
                               (const uint32_t*)e.leave_idx, e.hole_idx, e.fill_idx, (const uint32_t*)nullptr);
            hipLaunchKernelGGL(blubk::k_slab_migrate_fill, dim3(particle_blocks(n)), dim3(256), 0, G->stream, (const blubk::SlabCounts*)e.counts, (const uint32_t*)e.hole_idx,
                               (const uint32_t*)e.fill_idx, h->pos, h->pvel[0], h->pvel[1], h->pvel[2]);
        } else {
            if (has_up(G, i))
                hipLaunchKernelGGL(blubk::k_slab_select, dim3(particle_blocks(n)), dim3(256), 0, G->stream, n, (const float4*)h->pos, (const float4*)h->pvel[0], (const float4*)h->pvel[1],
                                   (const float4*)h->pvel[2], (float)h->slab_z1 - blubk::GHOST_MARGIN, (float)h->slab_z1, G->capacity, &e.counts->n_up, e.up[0],
                                   with_rows ? e.up[1] : nullptr, e.up[2], e.up[3], (const uint32_t*)nullptr);
            if (has_down(G, i))
                hipLaunchKernelGGL(blubk::k_slab_select, dim3(particle_blocks(n)), dim3(256), 0, G->stream, n, (const float4*)h->pos, (const float4*)h->pvel[0], (const float4*)h->pvel[1],
                                   (const float4*)h->pvel[2], (float)h->slab_z0, (float)h->slab_z0 + blubk::GHOST_MARGIN, G->capacity, &e.counts->n_down, e.dn[0],
                                   with_rows ? e.dn[1] : nullptr, e.dn[2], e.dn[3], (const uint32_t*)nullptr);
        }
    }
    // counts to the host (the payload sizes of the transfers below are host-side arguments)
    for (int i = 0; i < S; ++i) HIP_TRY(hipMemcpyAsync(&G->counts_host[i], G->ex[i].counts, sizeof(blubk::SlabCounts), hipMemcpyDeviceToHost, G->stream));
    G->comm_ops += 2;   // counts + payload
    if (G->rccl) {   // neighbours' counts
        NCCL_TRY(ncclGroupStart());
        for (int i = 0; i < S; ++i) {
            auto& e = G->ex[i];
            if (has_up(G, i) && !up_local(G, i)) { NCCL_TRY(ncclSend(&e.counts->n_up, 1, ncclUint32, G->first + i + 1, G->comm, G->stream)); NCCL_TRY(ncclRecv(e.recv_counts + 1, 1, ncclUint32, G->first + i + 1, G->comm, G->stream)); }
            if (has_down(G, i) && !down_local(G, i)) { NCCL_TRY(ncclSend(&e.counts->n_down, 1, ncclUint32, G->first + i - 1, G->comm, G->stream)); NCCL_TRY(ncclRecv(e.recv_counts + 0, 1, ncclUint32, G->first + i - 1, G->comm, G->stream)); }
        }
        NCCL_TRY(ncclGroupEnd());
        for (int i = 0; i < S; ++i) HIP_TRY(hipMemcpyAsync(&G->recv_host[2 * i], G->ex[i].recv_counts, 2 * sizeof(uint32_t), hipMemcpyDeviceToHost, G->stream));
    }
    HIP_TRY(hipStreamSynchronize(G->stream));
    G->host_syncs += 1;
    for (int i = 0; i < S; ++i)
        if (G->counts_host[i].n_up > G->capacity || G->counts_host[i].n_down > G->capacity) return set_error(BLUB_ERR_OUT_OF_MEMORY, "slab transfer buffer overflow");
    if (mode == XFER_MIGRATE)
        for (int i = 0; i < S; ++i) {
            blub_fluid* h = G->slabs[i];
            if (!h->num_particles) continue;
            h->num_particles = G->counts_host[i].n_stay;
        }
    // payload
    const int narr = with_rows ? 4 : 1;
    std::vector<uint32_t> from_below(S, 0), from_above(S, 0);
    for (int i = 0; i < S; ++i) {
        if (has_down(G, i)) from_below[i] = down_local(G, i) ? G->counts_host[i - 1].n_up : G->recv_host[2 * i + 0];
        if (has_up(G, i)) from_above[i] = up_local(G, i) ? G->counts_host[i + 1].n_down : G->recv_host[2 * i + 1];
        if ((uint64_t)G->slabs[i]->num_particles + from_below[i] + from_above[i] > G->capacity) return set_error(BLUB_ERR_OUT_OF_MEMORY, "slab particle capacity exceeded");
    }
    if (G->rccl) NCCL_TRY(ncclGroupStart());
    for (int i = 0; i < S; ++i) {
        blub_fluid* h = G->slabs[i];
        auto& e = G->ex[i];
        float4* dst[4] = {h->pos, h->pvel[0], h->pvel[1], h->pvel[2]};
        const size_t at_below = h->num_particles, at_above = (size_t)h->num_particles + from_below[i];
        for (int k = 0; k < narr; ++k) {
            if (has_down(G, i)) {
                if (down_local(G, i)) { if (from_below[i]) HIP_TRY(hipMemcpyAsync(dst[k] + at_below, G->ex[i - 1].up[k], (size_t)from_below[i] * 16, hipMemcpyDeviceToDevice, G->stream)); }
                else {
                    NCCL_TRY(ncclSend(e.dn[k], (size_t)G->counts_host[i].n_down * 16, ncclChar, G->first + i - 1, G->comm, G->stream));
                    NCCL_TRY(ncclRecv(dst[k] + at_below, (size_t)from_below[i] * 16, ncclChar, G->first + i - 1, G->comm, G->stream));
                }
            }
            if (has_up(G, i)) {
                if (up_local(G, i)) { if (from_above[i]) HIP_TRY(hipMemcpyAsync(dst[k] + at_above, G->ex[i + 1].dn[k], (size_t)from_above[i] * 16, hipMemcpyDeviceToDevice, G->stream)); }
                else {
                    NCCL_TRY(ncclSend(e.up[k], (size_t)G->counts_host[i].n_up * 16, ncclChar, G->first + i + 1, G->comm, G->stream));
                    NCCL_TRY(ncclRecv(dst[k] + at_above, (size_t)from_above[i] * 16, ncclChar, G->first + i + 1, G->comm, G->stream));
                }
            }
        }
    }
    if (G->rccl) NCCL_TRY(ncclGroupEnd());
    for (int i = 0; i < S; ++i) {
        blub_fluid* h = G->slabs[i];
        if (mode == XFER_MIGRATE) { h->num_particles += from_below[i] + from_above[i]; h->num_ghost = 0; }
        else h->num_ghost = from_below[i] + from_above[i];
        int rc = slab_upload_counts(G, i);
        if (rc != BLUB_OK) return rc;
        // what travelled: the next exchange of this kind sizes its messages from it (slab_exchange_particles_async)
        blub_slab_group::Hist& H = G->hist[(size_t)i * XFER_KINDS + kind];
        H.valid = true; H.pending = false; H.n_up = G->counts_host[i].n_up; H.n_down = G->counts_host[i].n_down; H.from_below = from_below[i]; H.from_above = from_above[i];
    }
    return BLUB_OK;
}

// Message capacity (particles) derived from the count that travelled over the same link in the same exchange of the previous step
static uint32_t slab_capx(const blub_slab_group* G, uint32_t prev) { return (uint32_t)std::min<uint64_t>(G->capacity, (((uint64_t)prev + prev / 2 + 2048u) + 255u) & ~255ull); }

// The exchange without host synchronisation (see blub_slab.hip.h): fixed-capacity messages with a count header, appended by a kernel.
static int slab_exchange_particles_async(blub_slab_group* G, int kind) {
    const int S = (int)G->slabs.size();
    const int mode = kind == XFER_MIGRATE_B ? XFER_MIGRATE : kind;
    const bool with_rows = mode != XFER_GHOST_POS, migrate = mode == XFER_MIGRATE;
    const int narr = with_rows ? 4 : 1;
    std::vector<uint32_t> cap_up(S, 0), cap_dn(S, 0), cap_below(S, 0), cap_above(S, 0);
    for (int i = 0; i < S; ++i) {      // what travelled last time: from the host (after a synchronous exchange) or from the pinned record the last append kernel wrote
        blub_slab_group::Hist& H = G->hist[(size_t)i * XFER_KINDS + kind];
        if (H.pending) {
            const volatile blubk::SlabXferRecord* r = &G->rec_host[((size_t)i * XFER_KINDS + kind) * XFER_RING + H.seq % XFER_RING];
            unsigned spins = 0;
            while (r->seq != H.seq) {      // (normally long since there: the record is a whole step old)
                if (++spins > 2000) { struct timespec ts = {0, 20000}; nanosleep(&ts, nullptr); }
                if (spins > 50000000u) return set_error(BLUB_ERR_DEVICE, "timed out waiting for the record of the previous particle exchange");
            }
            if (r->overflow & 1u) return set_error(BLUB_ERR_OUT_OF_MEMORY, "the particle capacity of a slab was exceeded by a particle exchange (max_num_particles is per slab)");
            if (r->overflow & 2u) return set_error(BLUB_ERR_COMM, "direct transport: a wait for a peer's flag timed out in an earlier step (a peer stopped stepping or fell seconds behind); "
                                                                  "what was stepped since is invalid -- blub_slab_group_synchronize reports and clears the condition");
            H.n_up = r->n_up; H.n_down = r->n_down; H.from_below = r->from_below; H.from_above = r->from_above; H.pending = false;
        }
        cap_up[i] = has_up(G, i) ? slab_capx(G, H.n_up) : 0u; cap_dn[i] = has_down(G, i) ? slab_capx(G, H.n_down) : 0u;
        cap_below[i] = has_down(G, i) ? slab_capx(G, H.from_below) : 0u; cap_above[i] = has_up(G, i) ? slab_capx(G, H.from_above) : 0u;
        if (G->direct) {      // the sender writes what travels straight into staging buffers of full capacity: no message size, nothing to hold back
            cap_up[i] = has_up(G, i) ? G->capacity : 0u; cap_dn[i] = has_down(G, i) ? G->capacity : 0u;
            cap_below[i] = has_down(G, i) ? G->capacity : 0u; cap_above[i] = has_up(G, i) ? G->capacity : 0u;
        }
    }
    G->xfer_seq += 1;
    for (int i = 0; i < S; ++i) {
        blub_fluid* h = G->slabs[i];
        auto& e = G->ex[i];
        h->num_particles = G->capacity;      // from here on: bounds (the counts live in n_dev)
        HIP_TRY(hipMemsetAsync(e.counts, 0, sizeof(blubk::SlabCounts), G->stream));
        const uint32_t n = h->num_particles;
        const uint32_t* nd = h->n_dev;
        if (migrate) {
            hipLaunchKernelGGL(blubk::k_slab_migrate_mark, dim3(particle_blocks(n)), dim3(256), 0, G->stream, n, (const float4*)h->pos, (const float4*)h->pvel[0], (const float4*)h->pvel[1],
                               (const float4*)h->pvel[2], (float)h->slab_z0, (float)h->slab_z1, G->capacity, e.counts, e.up[0], e.up[1], e.up[2], e.up[3],
                               e.dn[0], e.dn[1], e.dn[2], e.dn[3], e.leave_idx, nd, cap_up[i], cap_dn[i], h->pos);
            hipLaunchKernelGGL(blubk::k_slab_migrate_match, dim3(particle_blocks(n)), dim3(256), 0, G->stream, n, (const float4*)h->pos, (float)h->slab_z0, (float)h->slab_z1, e.counts,
                               (const uint32_t*)e.leave_idx, e.hole_idx, e.fill_idx, nd);
            hipLaunchKernelGGL(blubk::k_slab_migrate_fill, dim3(particle_blocks(n)), dim3(256), 0, G->stream, (const blubk::SlabCounts*)e.counts, (const uint32_t*)e.hole_idx,
                               (const uint32_t*)e.fill_idx, h->pos, h->pvel[0], h->pvel[1], h->pvel[2]);
        } else {
            if (has_up(G, i))
                hipLaunchKernelGGL(blubk::k_slab_select, dim3(particle_blocks(n)), dim3(256), 0, G->stream, n, (const float4*)h->pos, (const float4*)h->pvel[0], (const float4*)h->pvel[1],
                                   (const float4*)h->pvel[2], (float)h->slab_z1 - blubk::GHOST_MARGIN, (float)h->slab_z1, G->capacity, &e.counts->n_up, e.up[0],
                                   with_rows ? e.up[1] : nullptr, e.up[2], e.up[3], nd);
            if (has_down(G, i))
                hipLaunchKernelGGL(blubk::k_slab_select, dim3(particle_blocks(n)), dim3(256), 0, G->stream, n, (const float4*)h->pos, (const float4*)h->pvel[0], (const float4*)h->pvel[1],
                                   (const float4*)h->pvel[2], (float)h->slab_z0, (float)h->slab_z0 + blubk::GHOST_MARGIN, G->capacity, &e.counts->n_down, e.dn[0],
                                   with_rows ? e.dn[1] : nullptr, e.dn[2], e.dn[3], nd);
        }
        hipLaunchKernelGGL(blubk::k_slab_finish_send, dim3(1), dim3(1), 0, G->stream, (const blubk::SlabCounts*)e.counts, has_up(G, i) ? e.up_msg : (float4*)nullptr,
                           has_down(G, i) ? e.dn_msg : (float4*)nullptr, cap_up[i], cap_dn[i], h->n_dev, (int)migrate);
    }
    // transport: ONE grouped operation (round 2: counts, a host synchronisation, then the payload)
    G->comm_ops += 1;
    if (G->direct) {
        G->flag_seq = seq_after(G->flag_seq);
        for (int i = 0; i < S; ++i) {
            auto& e = G->ex[i];
            blubk::SlabParticlePush up{}, dn{};
            blubk::SlabFlagList F{}; F.seq = G->flag_seq; F.blocks_done = G->blocks_done[i];
            for (int k = 0; k < narr; ++k) {
                up.src[k] = k == 0 ? e.up_msg : e.up[k]; dn.src[k] = k == 0 ? e.dn_msg : e.dn[k];
                // my "up" message is the upper neighbour's "from below" staging, my "down" message the lower neighbour's "from above"
                up.dst[k] = has_up(G, i) ? peer_ptr(G, i, G->first + i + 1, k == 0 ? e.rb_msg : e.rb[k]) : nullptr;
                dn.dst[k] = has_down(G, i) ? peer_ptr(G, i, G->first + i - 1, k == 0 ? e.ra_msg : e.ra[k]) : nullptr;
            }
            if (has_up(G, i)) flag_list_add(G, F, i, G->first + i + 1);
            if (has_down(G, i)) flag_list_add(G, F, i, G->first + i - 1);
            if (F.n) hipLaunchKernelGGL(blubk::k_slab_push_particles, dim3(64, 2), dim3(256), 0, G->stream, up, dn, narr, F);
        }
        for (int i = 0; i < S; ++i) { int rc = slab_wait(G, i, neighbour_mask(G, i)); if (rc != BLUB_OK) return rc; }
    }
    if (G->rccl && !G->direct) NCCL_TRY(ncclGroupStart());
    for (int i = 0; i < S && !G->direct; ++i) {
        auto& e = G->ex[i];
        for (int k = 0; k < narr; ++k) {
            const size_t hdr = k == 0 ? 1 : 0;      // the position message carries the header
            const float4* up_src = k == 0 ? e.up_msg : e.up[k]; const float4* dn_src = k == 0 ? e.dn_msg : e.dn[k];
            float4* ra_dst = k == 0 ? e.ra_msg : e.ra[k]; float4* rb_dst = k == 0 ? e.rb_msg : e.rb[k];
            if (has_up(G, i)) {
                if (up_local(G, i)) {      // my up message is the next slab's "from below", its down message my "from above"
                    auto& en = G->ex[i + 1];
                    { int rc = slab_copy(G, k == 0 ? en.rb_msg : en.rb[k], up_src, (hdr + cap_up[i]) * 16); if (rc != BLUB_OK) return rc; }
                    { int rc = slab_copy(G, ra_dst, k == 0 ? en.dn_msg : en.dn[k], (hdr + cap_above[i]) * 16); if (rc != BLUB_OK) return rc; }
                } else {
                    NCCL_TRY(ncclSend(up_src, (hdr + cap_up[i]) * 16, ncclChar, G->first + i + 1, G->comm, G->stream));
                    NCCL_TRY(ncclRecv(ra_dst, (hdr + cap_above[i]) * 16, ncclChar, G->first + i + 1, G->comm, G->stream));
                }
            }
            if (has_down(G, i) && !down_local(G, i)) {
                NCCL_TRY(ncclSend(dn_src, (hdr + cap_dn[i]) * 16, ncclChar, G->first + i - 1, G->comm, G->stream));
                NCCL_TRY(ncclRecv(rb_dst, (hdr + cap_below[i]) * 16, ncclChar, G->first + i - 1, G->comm, G->stream));
            }
        }
    }
    if (G->rccl && !G->direct) NCCL_TRY(ncclGroupEnd());
    if (!G->direct) { int rc = slab_copy_flush(G); if (rc != BLUB_OK) return rc; }
    for (int i = 0; i < S; ++i) {
        blub_fluid* h = G->slabs[i];
        auto& e = G->ex[i];
        blubk::SlabAppendArgs a{};
        for (int k = 0; k < 4; ++k) { a.below[k] = has_down(G, i) ? (k == 0 ? e.rb_msg : e.rb[k]) : nullptr; a.above[k] = has_up(G, i) ? (k == 0 ? e.ra_msg : e.ra[k]) : nullptr; }
        a.dst[0] = h->pos; a.dst[1] = h->pvel[0]; a.dst[2] = h->pvel[1]; a.dst[3] = h->pvel[2];
        blubk::SlabXferRecord* rec = G->rec_dev + ((size_t)i * XFER_KINDS + kind) * XFER_RING + G->xfer_seq % XFER_RING;
        // (<= 64 workgroups, grid-stride: every workgroup ends with an atomic on ONE counter -- 512 of them took 11-13 us to append a few thousand records)
        hipLaunchKernelGGL(blubk::k_slab_append, dim3(std::max(1u, std::min(64u, particle_blocks(cap_below[i] + cap_above[i])))), dim3(256), 0, G->stream, a, narr, cap_below[i], cap_above[i], G->capacity,
                           h->n_dev, (int)migrate, (const blubk::SlabCounts*)e.counts, rec, G->xfer_seq, e.append_done, G->direct ? (const uint32_t*)G->dir_error[i] : (const uint32_t*)nullptr);
        blub_slab_group::Hist& H = G->hist[(size_t)i * XFER_KINDS + kind];
        H.pending = true; H.seq = G->xfer_seq;
        h->num_ghost = migrate ? 0u : cap_below[i] + cap_above[i];      // (bound)
    }
    return BLUB_OK;
}

// exact particle counts of every local slab back on the host (blocks): for the entry points that hand particles out
static int slab_refresh_counts(blub_slab_group* G) {
    HIP_TRY(hipStreamSynchronize(G->stream));
    for (size_t i = 0; i < G->slabs.size(); ++i) {
        uint32_t v[3] = {0, 0, 0};
        HIP_TRY(hipMemcpy(v, G->slabs[i]->n_dev, sizeof v, hipMemcpyDeviceToHost));
        if (v[2]) return set_error(BLUB_ERR_OUT_OF_MEMORY, "the particle capacity of a slab was exceeded by a particle exchange (max_num_particles is per slab)");
        G->slabs[i]->num_particles = v[0]; G->slabs[i]->num_ghost = v[1];
    }
    return BLUB_OK;
}

// PressureSolver::solve on all slabs in lock step (brick mapping; the blub_pcg.hip.h kernels fed with the gathered partials).
//
// Iteration count: a solve that converges at check iteration c sets `done` in the direction kernel of iteration c+1 and
// everything launched after that is a no-op -- but the transport operations between the kernels would still be paid.  So the
// host launches iterations through the check that ended the PREVIOUS solve of this kind, synchronises, looks at `done`, and
// adds one check interval at a time until the solve is finished.  Every rank sees bit-identical control blocks (same
// partials, same reduction order), and the previous iteration count is only taken from a read-back that a stream
// synchronisation has completed on every rank, so all ranks issue the same sequence of transport operations.
static int slab_solve_single_reduction(blub_slab_group* G, int which, float dt);
static int slab_solve(blub_slab_group* G, int which, float dt) {
    const int S = (int)G->slabs.size();
    blub_fluid* h0 = G->slabs[0];
    if (h0->precond_mode != BLUB_PRECOND_ZERO) return set_error(BLUB_ERR_UNSUPPORTED, "z-slab groups support the default preconditioner reading only");
    const bool single = h0->pcg_schedule == 1 && h0->cfg[which].max_num_iterations <= h0->pcg1_max_iterations;
    for (auto h : G->slabs) { h->last_schedule[which] = single ? 1 : 0; h->last_mapping[which] = 1; }      // (slab solves always run on the brick mapping)
    if (single) return slab_solve_single_reduction(G, which, dt);
    const blub_solver_config c = h0->cfg[which];
    const float tol = c.error_tolerance / dt;
    const int maxit = c.max_num_iterations, freq = c.error_check_frequency;
    auto is_check = [&](int j) { return j > 0 && freq > 0 && j % freq == 0; };
    const int np = G->np_cur, npall = np * G->nranks;
    const dim3 grid(np), block(PCG_B_THREADS);
    int rc;
    auto seg_upd = [&](int i) { return G->ex[i].gat_upd + (size_t)(G->first + i) * np; };
    auto seg_dir = [&](int i) { return G->ex[i].gat_dir + (size_t)(G->first + i) * np; };
    auto gather_upd = [&](bool own_group) { return slab_gather(G, [G](int i) { return reinterpret_cast<float*>(G->ex[i].gat_upd); }, 2 * np, own_group); };
    auto gather_dir = [&]() { return slab_gather(G, [G](int i) { return G->ex[i].gat_dir; }, np); };
    // iterations to launch before the first look at `done`
    int target = maxit + 1;
    if (freq > 0 && G->ctrl_host_valid[which]) {
        const int prev = (int)G->ctrl_host[which].num_iter;
        if (prev >= 0 && prev < maxit) target = std::min(maxit + 1, (prev / freq) * freq + 2);
    }
    for (int i = 0; i < S; ++i) {
        blub_fluid* h = G->slabs[i];
        if (!h->pressure_initialised[which]) { int rz = vol_zero(h, h->pressure[which]); if (rz != BLUB_OK) return rz; h->pressure_initialised[which] = true; }
        h->solve_seq[which] += 1;
        LAUNCH(h, KC_PCG_INIT, k_pcg_init_b<false>, grid, block, h->bg, LIST(h, active), (const uint32_t*)&h->counts->n_fluid, np, (const int8_t*)h->marker, h->dvol, h->pressure[which], h->residual, h->search,
               seg_upd(i), h->ctrl[which], (PcgTailSync*)nullptr, DivergenceSrc{});
    }
    // descriptor, r, s planes and the initial partials: one grouped operation
    if ((rc = slab_fused(G, [&]() -> int {
            int r2 = slab_halo(G, {[](blub_fluid* h) { return (void*)h->dvol; }}, 1, false);
            return r2 != BLUB_OK ? r2 : slab_halo(G, {[](blub_fluid* h) { return (void*)h->residual; }, [](blub_fluid* h) { return (void*)h->search; }}, 4, false);
        }, gather_upd)) != BLUB_OK) return rc;
    int it = 0;
    bool done = false;
    for (;;) {
        for (; it < target; ++it) {
            for (int i = 0; i < S; ++i) {
                blub_fluid* h = G->slabs[i];
                float* sbuf[2] = {h->search, h->aux};
                const int halo_lo = has_down(G, i) ? h->slab_z0 : -1, halo_hi = has_up(G, i) ? h->slab_z1 - 1 : -1;
                if (it == 0)
                    LAUNCH(h, KC_PCG_DIR, (k_pcg_dir_s<true, true>), grid, block, h->bg, LIST(h, fluid), np, (const uint8_t*)h->dvol, (const float*)h->residual, (const float*)sbuf[0], sbuf[0],
                           (const float2*)G->ex[i].gat_upd, seg_dir(i), npall, h->ctrl[which], tol, it, 0, halo_lo, halo_hi);
                else
                    LAUNCH(h, KC_PCG_DIR, (k_pcg_dir_s<false, true>), grid, block, h->bg, LIST(h, fluid), np, (const uint8_t*)h->dvol, (const float*)h->residual, (const float*)sbuf[(it - 1) & 1], sbuf[it & 1],
                           (const float2*)G->ex[i].gat_upd, seg_dir(i), npall, h->ctrl[which], tol, it, (int)is_check(it - 1), halo_lo, halo_hi);
            }
            if ((rc = gather_dir()) != BLUB_OK) return rc;
            for (int i = 0; i < S; ++i) {
                blub_fluid* h = G->slabs[i];
                float* sbuf[2] = {h->search, h->aux};
                LAUNCH(h, KC_PCG_UPDATE, k_pcg_update_s, grid, block, h->bg, LIST(h, fluid), np, (const uint8_t*)h->dvol, (const float*)sbuf[it & 1], h->pressure[which], h->residual,
                       (const float*)G->ex[i].gat_dir, seg_upd(i), npall, (const PcgCtrl*)h->ctrl[which], it);
            }
            if ((rc = slab_fused(G, [&]() -> int { return slab_halo(G, {[](blub_fluid* h) { return (void*)h->residual; }}, 4, false); }, gather_upd)) != BLUB_OK) return rc;
        }
        if (target > maxit) break;
        HIP_TRY(hipMemcpyAsync(&G->ctrl_host[which], h0->ctrl[which], sizeof(PcgCtrl), hipMemcpyDeviceToHost, G->stream));
        HIP_TRY(hipStreamSynchronize(G->stream));
        G->done_polls += 1;
        done = G->ctrl_host[which].done != 0;
        if (done) break;
        target = std::min(maxit + 1, target + freq);
    }
    for (int i = 0; i < S; ++i) {
        blub_fluid* h = G->slabs[i];
        LAUNCH(h, KC_PCG_FINALIZE, k_pcg_finalize, dim3(1), dim3(256), h->ctrl[which], (const float2*)G->ex[i].gat_upd, npall, (const uint32_t*)nullptr, maxit, h->solve_seq[which], (PcgCtrl*)nullptr);
        if (maxit & 1) std::swap(h->search, h->aux);
        if ((rc = enqueue_stats_readback(h, which, dt)) != BLUB_OK) return rc;
    }
    // iteration count of this solve for the next one: lands before the next particle exchange synchronises the stream
    HIP_TRY(hipMemcpyAsync(&G->ctrl_host[which], h0->ctrl[which], sizeof(PcgCtrl), hipMemcpyDeviceToHost, G->stream));
    G->ctrl_host_valid[which] = true;
    const int w = which;
    return slab_halo(G, {[w](blub_fluid* h) { return (void*)h->pressure[w]; }}, 4);
}

// The same solve with the single-reduction schedule (blub_pcg1.hip.h): ONE kernel and ONE grouped transport operation per iteration
// -- the w = A M^-1 r plane to the z-neighbours together with the {gamma, delta, max|r|} partials to every slab.  r and q of the
// ghost planes are recomputed locally (HALO variant of k_pcg1_iter_s), exactly like s in the two-kernel schedule.
static int slab_solve_single_reduction(blub_slab_group* G, int which, float dt) {
    const int S = (int)G->slabs.size();
    blub_fluid* h0 = G->slabs[0];
    const blub_solver_config c = h0->cfg[which];
    const float tol = c.error_tolerance / dt;
    const int maxit = c.max_num_iterations, freq = c.error_check_frequency;
    auto is_check = [&](int j) { return j > 0 && freq > 0 && j % freq == 0; };
    const int np = G->np_cur, npall = np * G->nranks;
    const dim3 grid(np), block(PCG_B_THREADS);
    int rc;
    auto seg_upd = [&](int i) { return G->ex[i].gat_upd + (size_t)(G->first + i) * np; };
    auto seg4 = [&](int i, int par) { return G->ex[i].gat4[par] + (size_t)(G->first + i) * np; };
    auto gather_upd = [&](bool own_group) { return slab_gather(G, [G](int i) { return reinterpret_cast<float*>(G->ex[i].gat_upd); }, 2 * np, own_group); };
    int target = maxit + 1;
    if (freq > 0 && G->ctrl_host_valid[which]) {
        const int prev = (int)G->ctrl_host[which].num_iter;
        if (prev >= 0 && prev < maxit) target = std::min(maxit + 1, (prev / freq) * freq + 2);
    }
    for (int i = 0; i < S; ++i) {
        blub_fluid* h = G->slabs[i];
        if ((rc = ensure_pcg1_buffers(h)) != BLUB_OK) return rc;
        if (!h->pressure_initialised[which]) { int rz = vol_zero(h, h->pressure[which]); if (rz != BLUB_OK) return rz; h->pressure_initialised[which] = true; }
        h->solve_seq[which] += 1;
        LAUNCH(h, KC_PCG_INIT, k_pcg_init_b<false>, grid, block, h->bg, LIST(h, active), (const uint32_t*)&h->counts->n_fluid, np, (const int8_t*)h->marker, h->dvol, h->pressure[which], h->residual, h->search,
               seg_upd(i), h->ctrl[which], (PcgTailSync*)nullptr, DivergenceSrc{});
    }
    // descriptor, r_0 and u_0 = M^-1 r_0 planes and the gamma_0 partials: one grouped operation
    if ((rc = slab_fused(G, [&]() -> int {
            int r2 = slab_halo(G, {[](blub_fluid* h) { return (void*)h->dvol; }}, 1, false);
            return r2 != BLUB_OK ? r2 : slab_halo(G, {[](blub_fluid* h) { return (void*)h->residual; }, [](blub_fluid* h) { return (void*)h->search; }}, 4, false);
        }, gather_upd)) != BLUB_OK) return rc;
    for (auto h : G->slabs) if (h->scalar_log[which]) HIP_TRY(hipMemsetAsync(h->scalar_log[which], 0xFF, 1024 * sizeof(float4), G->stream));
    // per-slab buffer roles of this solve (see stage_solve)
    struct Bufs { float* R[2]; float* W[2]; float* Q[2]; };
    std::vector<Bufs> B(S);
    for (int i = 0; i < S; ++i) {
        blub_fluid* h = G->slabs[i];
        B[i] = Bufs{{h->residual, h->cgbuf[0]}, {h->aux, h->cgbuf[1]}, {h->aux_temp, h->cgbuf[2]}};
        if (G->direct) {
            // direct transport: the kernel pushes its boundary planes of w_0 and its partials itself, tagged with the number K(0) expects
            const int halo_lo = has_down(G, i) ? h->slab_z0 : -1, halo_hi = has_up(G, i) ? h->slab_z1 - 1 : -1;
            SlabDirect D{};
            const int me = G->first + i;
            D.w_up = has_up(G, i) ? peer_vol(G, i, me + 1, B[i].W[0]) : nullptr; D.w_dn = has_down(G, i) ? peer_vol(G, i, me - 1, B[i].W[0]) : nullptr;
            for (int r = 0; r < G->nranks; ++r) {
                if (r == me) continue;
                D.part_out[D.n_out] = peer_ptr(G, i, r, seg4(i, 0)); D.n_out += 1;
            }
            LAUNCH(h, KC_PCG_INIT, k_pcg1_w0_s<true>, grid, block, h->bg, LIST(h, fluid), np, (const uint8_t*)h->dvol, (const float*)h->search, B[i].W[0], (const float2*)G->ex[i].gat_upd, npall,
                   seg4(i, 0), (int)(G->first + i == 0), seq_after(G->flag_seq), halo_lo, halo_hi, D);
        } else
            LAUNCH(h, KC_PCG_INIT, k_pcg1_w0_s<false>, grid, block, h->bg, LIST(h, fluid), np, (const uint8_t*)h->dvol, (const float*)h->search, B[i].W[0], (const float2*)G->ex[i].gat_upd, npall,
                   seg4(i, 0), (int)(G->first + i == 0), 0u, -1, -1, SlabDirect{});
    }
    auto exchange = [&](int wpar, int ppar) -> int {   // plane of W[wpar] to the z-neighbours + partials of parity ppar to every slab
        return slab_fused(G, [&]() -> int {
            // (the volumes differ per slab: pass the slab index through a per-call table)
            const blub_fluid* h0l = G->slabs[0];
            const size_t pb = (size_t)h0l->g.nx * h0l->g.ny * sizeof(float);
            for (int i = 0; i < S && G->direct; ++i) {      // direct transport: push the two boundary planes into the z-neighbours' copies
                blub_fluid* h = G->slabs[i];
                char* base = (char*)B[i].W[wpar];
                if (has_up(G, i)) { int r3 = slab_copy(G, peer_vol(G, i, G->first + i + 1, base, sizeof(float)) + (size_t)(h->slab_z1 - 1) * pb, base + (size_t)(h->slab_z1 - 1) * pb, pb); if (r3 != BLUB_OK) return r3; push_flag_to(G, i, G->first + i + 1); }
                if (has_down(G, i)) { int r3 = slab_copy(G, peer_vol(G, i, G->first + i - 1, base, sizeof(float)) + (size_t)h->slab_z0 * pb, base + (size_t)h->slab_z0 * pb, pb); if (r3 != BLUB_OK) return r3; push_flag_to(G, i, G->first + i - 1); }
            }
            for (int i = 0; i < S && !G->direct; ++i) {
                blub_fluid* h = G->slabs[i];
                char* base = (char*)B[i].W[wpar];
                if (has_up(G, i)) {
                    if (up_local(G, i)) {
                        char* nb = (char*)B[i + 1].W[wpar];
                        { int r3 = slab_copy(G, nb + (size_t)(h->slab_z1 - 1) * pb, base + (size_t)(h->slab_z1 - 1) * pb, pb); if (r3 != BLUB_OK) return r3; }
                        { int r3 = slab_copy(G, base + (size_t)h->slab_z1 * pb, nb + (size_t)h->slab_z1 * pb, pb); if (r3 != BLUB_OK) return r3; }
                    } else {
                        NCCL_TRY(ncclSend(base + (size_t)(h->slab_z1 - 1) * pb, pb, ncclChar, G->first + i + 1, G->comm, G->stream));
                        NCCL_TRY(ncclRecv(base + (size_t)h->slab_z1 * pb, pb, ncclChar, G->first + i + 1, G->comm, G->stream));
                    }
                }
                if (has_down(G, i) && !down_local(G, i)) {
                    NCCL_TRY(ncclSend(base + (size_t)h->slab_z0 * pb, pb, ncclChar, G->first + i - 1, G->comm, G->stream));
                    NCCL_TRY(ncclRecv(base + (size_t)(h->slab_z0 - 1) * pb, pb, ncclChar, G->first + i - 1, G->comm, G->stream));
                }
            }
            return BLUB_OK;
        }, [&](bool own_group) { return slab_gather(G, [G, ppar](int i) { return reinterpret_cast<float*>(G->ex[i].gat4[ppar]); }, 4 * np, own_group); });
    };
    if (G->direct) G->flag_seq = seq_after(G->flag_seq);      // (the number the w_0 kernels tagged their partials with: no exchange of its own)
    else if ((rc = exchange(0, 0)) != BLUB_OK) return rc;
    int it = 0;
    if (G->direct) {
        // DIRECT transport: K(i) itself stores its boundary planes of w_{i+1} and of p into the z-neighbours' ghost planes and its partials into
        // every slab's array, then raises their flags; K(i + 1) waits for the flags it needs in its prologue.  No host-issued operation between
        // the launches, no look at `done`: every iteration up to the cap is launched, a finished solve's launches return at once (they neither
        // wait nor publish -- every slab takes that decision alike, from bit-identical scalars).
        for (it = 0; it <= maxit; ++it) {
            const uint32_t seq_in = G->flag_seq;
            G->flag_seq = seq_after(G->flag_seq);
            for (int i = 0; i < S; ++i) {
                blub_fluid* h = G->slabs[i];
                const int halo_lo = has_down(G, i) ? h->slab_z0 : -1, halo_hi = has_up(G, i) ? h->slab_z1 - 1 : -1;
                const float4* pin = G->ex[i].gat4[it & 1];
                float4* pout = seg4(i, (it + 1) & 1);
                float* wout = B[i].W[(it + 1) & 1];
                SlabDirect D{};
                const int me = G->first + i;
                D.w_up = has_up(G, i) ? peer_vol(G, i, me + 1, wout) : nullptr; D.w_dn = has_down(G, i) ? peer_vol(G, i, me - 1, wout) : nullptr;
                D.p_up = has_up(G, i) ? peer_vol(G, i, me + 1, h->pressure[which]) : nullptr; D.p_dn = has_down(G, i) ? peer_vol(G, i, me - 1, h->pressure[which]) : nullptr;
                for (int r = 0; r < G->nranks; ++r) {
                    if (r == me) continue;
                    D.part_out[D.n_out] = peer_ptr(G, i, r, pout); D.flag_out[D.n_out] = flag_of(G, i, r); D.n_out += 1;
                }
                D.flags_in = G->flags[i]; D.blocks_done = G->blocks_done[i]; D.error = G->dir_error[i];
                D.seq_in = seq_in; D.seq_out = G->flag_seq; D.wait_mask = others_mask(G, i); D.log = h->scalar_log[which];
                if (it == 0)
                    LAUNCH(h, KC_PCG_ITER, (k_pcg1_iter_s<true, true, true, true>), grid, block, h->bg, LIST(h, fluid), np, (const uint8_t*)h->dvol, (const float*)B[i].R[0], B[i].R[1], (const float*)B[i].W[0],
                           B[i].W[1], (const float*)B[i].Q[1], B[i].Q[0], h->search, h->pressure[which], pin, pout, npall, h->ctrl[which], h->pcg1_scalars[which], tol, 0, 0, halo_lo, halo_hi, D);
                else
                    LAUNCH(h, KC_PCG_ITER, (k_pcg1_iter_s<false, true, true, true>), grid, block, h->bg, LIST(h, fluid), np, (const uint8_t*)h->dvol, (const float*)B[i].R[it & 1], B[i].R[(it + 1) & 1],
                           (const float*)B[i].W[it & 1], B[i].W[(it + 1) & 1], (const float*)B[i].Q[(it + 1) & 1], B[i].Q[it & 1], h->search, h->pressure[which], pin, pout, npall,
                           h->ctrl[which], h->pcg1_scalars[which], tol, it, (int)is_check(it - 1), halo_lo, halo_hi, D);
            }
        }
        for (int i = 0; i < S; ++i) {
            blub_fluid* h = G->slabs[i];
            // (a solve that ran into the iteration cap takes its statistics from the partials of K(maxit): accepted by their tags unless it is over)
            LAUNCH(h, KC_PCG_FINALIZE, k_pcg1_finalize, dim3(1), dim3(256), h->ctrl[which], (const float4*)G->ex[i].gat4[(maxit + 1) & 1], npall, (const uint32_t*)nullptr, maxit, h->solve_seq[which], (PcgCtrl*)nullptr,
                   G->flag_seq, G->dir_error[i]);
            if ((maxit + 1) & 1) std::swap(h->residual, h->cgbuf[0]);
            if ((rc = enqueue_stats_readback(h, which, dt)) != BLUB_OK) return rc;
        }
        return BLUB_OK;      // (the pressure halo is current: every launched iteration pushed its boundary planes of p)
    }
    for (;;) {
        for (; it < target; ++it) {
            for (int i = 0; i < S; ++i) {
                blub_fluid* h = G->slabs[i];
                const int halo_lo = has_down(G, i) ? h->slab_z0 : -1, halo_hi = has_up(G, i) ? h->slab_z1 - 1 : -1;
                const float4* pin = G->ex[i].gat4[it & 1];
                float4* pout = seg4(i, (it + 1) & 1);
                SlabDirect Dlog{}; Dlog.log = h->scalar_log[which];
                if (it == 0)
                    LAUNCH(h, KC_PCG_ITER, (k_pcg1_iter_s<true, true>), grid, block, h->bg, LIST(h, fluid), np, (const uint8_t*)h->dvol, (const float*)B[i].R[0], B[i].R[1], (const float*)B[i].W[0],
                           B[i].W[1], (const float*)B[i].Q[1], B[i].Q[0], h->search, h->pressure[which], pin, pout, npall, h->ctrl[which], h->pcg1_scalars[which], tol, 0, 0, halo_lo, halo_hi, Dlog);
                else
                    LAUNCH(h, KC_PCG_ITER, (k_pcg1_iter_s<false, true>), grid, block, h->bg, LIST(h, fluid), np, (const uint8_t*)h->dvol, (const float*)B[i].R[it & 1], B[i].R[(it + 1) & 1],
                           (const float*)B[i].W[it & 1], B[i].W[(it + 1) & 1], (const float*)B[i].Q[(it + 1) & 1], B[i].Q[it & 1], h->search, h->pressure[which], pin, pout, npall,
                           h->ctrl[which], h->pcg1_scalars[which], tol, it, (int)is_check(it - 1), halo_lo, halo_hi, Dlog);
            }
            if ((rc = exchange((it + 1) & 1, (it + 1) & 1)) != BLUB_OK) return rc;
        }
        if (target > maxit) break;
        HIP_TRY(hipMemcpyAsync(&G->ctrl_host[which], h0->ctrl[which], sizeof(PcgCtrl), hipMemcpyDeviceToHost, G->stream));
        HIP_TRY(hipStreamSynchronize(G->stream));
        G->done_polls += 1;
        if (G->ctrl_host[which].done != 0) break;
        target = std::min(maxit + 1, target + freq);
    }
    for (int i = 0; i < S; ++i) {
        blub_fluid* h = G->slabs[i];
        LAUNCH(h, KC_PCG_FINALIZE, k_pcg1_finalize, dim3(1), dim3(256), h->ctrl[which], (const float4*)G->ex[i].gat4[(maxit + 1) & 1], npall, (const uint32_t*)nullptr, maxit, h->solve_seq[which], (PcgCtrl*)nullptr,
               0u, (uint32_t*)nullptr);
        if ((maxit + 1) & 1) std::swap(h->residual, h->cgbuf[0]);
        if ((rc = enqueue_stats_readback(h, which, dt)) != BLUB_OK) return rc;
    }
    HIP_TRY(hipMemcpyAsync(&G->ctrl_host[which], h0->ctrl[which], sizeof(PcgCtrl), hipMemcpyDeviceToHost, G->stream));
    G->ctrl_host_valid[which] = true;
    const int w = which;
    return slab_halo(G, {[w](blub_fluid* h) { return (void*)h->pressure[w]; }}, 4);
}

// HybridFluid::step (hybrid_fluid.rs:770-977) over all slabs in lock step
// ghosts are dropped: on the host (bound) and on the device (count)
static int slab_drop_ghosts(blub_slab_group* G) {
    for (auto h : G->slabs) { h->num_ghost = 0; HIP_TRY(hipMemsetAsync(h->n_dev + 1, 0, sizeof(uint32_t), G->stream)); }
    return BLUB_OK;
}

// HybridFluid::step (hybrid_fluid.rs:770-977) over all slabs in lock step.  The step is cut into the segments below so that a test can
// stop between them and look at every slab (blub_slab_group_run_stages); blub_slab_group_step runs all of them.
enum SlabStage { SS_GHOSTS = 0, SS_TRANSFER, SS_DIVERGENCE, SS_SOLVE_VELOCITY, SS_BINNING, SS_PROJECT, SS_ADVECT, SS_MIGRATE, SS_DENSITY_GATHER,
                 SS_SOLVE_DENSITY, SS_POSITION_CHANGE, SS_CORRECT, SS_MIGRATE_B, SS_FINISH, SS_COUNT };
static int slab_step(blub_slab_group* G, float dt, int first = 0, int last = SS_FINISH) {
    int rc;
    const int S = (int)G->slabs.size();
    blub_fluid* h0 = G->slabs[0];
#define FOR_SLABS(call) for (int i = 0; i < S; ++i) { blub_fluid* h = G->slabs[i]; (void)h; if ((rc = (call)) != BLUB_OK) return rc; }
#define RUNS(stage) (first <= (stage) && (stage) <= last)
    // The particle exchanges run without host synchronisation once every exchange kind has a history to size its messages from (i.e. from
    // the second step after the particles were set); all ranks take the same decision (they step in lock step).
    bool async = G->async_exchange;
    for (auto& H : G->hist) async = async && (H.valid || G->direct);      // (direct transport: no message sizes, hence no history needed)
    auto exchange = [&](int kind) { return async ? slab_exchange_particles_async(G, kind) : slab_exchange_particles(G, kind); };
    if (RUNS(SS_GHOSTS)) {
        // size of this step's PCG grids: every slab contributes the newest fluid-brick count it has (a lagged, non-blocking snapshot) and
        // all of them use the same grid, from the largest count (one 4-byte gather per step).  The gathered values reach the host through a
        // pinned, tagged record and are used by the NEXT step (asynchronous path) / after the sync of the exchange below (synchronous path)
        if (async && G->cnt_pending) {
            const volatile uint32_t* tag = &G->cntseq_host[G->cnt_seq & 1];
            unsigned spins = 0;
            while (*tag != G->cnt_seq) { if (++spins > 2000) { struct timespec ts = {0, 20000}; nanosleep(&ts, nullptr); } if (spins > 50000000u) return set_error(BLUB_ERR_DEVICE, "timed out waiting for the gathered brick counts"); }
            for (int k = 0; k < G->nranks; ++k) G->cnt_host[k] = G->cntrec_host[(size_t)(G->cnt_seq & 1) * G->nranks + k];
        }
        for (int i = 0; i < S; ++i) {
            BrickCounts bc{}; bool have = false;
            if ((rc = latest_counts(G->slabs[i], false, &bc, &have)) != BLUB_OK) return rc;
            G->cnt_host[G->nranks + i] = have ? (float)bc.n_fluid : 0.0f;
            HIP_TRY(hipMemcpyAsync(G->ex[i].gat_cnt + (G->first + i), &G->cnt_host[G->nranks + i], sizeof(float), hipMemcpyHostToDevice, G->stream));
        }
        if ((rc = slab_gather(G, [G](int i) { return G->ex[i].gat_cnt; }, 1)) != BLUB_OK) return rc;
        if (async) {
            G->cnt_seq += 1; G->cnt_pending = true;
            hipLaunchKernelGGL(k_slab_publish_cnt, dim3(1), dim3(1), 0, G->stream, (const float*)G->ex[0].gat_cnt, G->nranks, G->cntrec_dev + (size_t)(G->cnt_seq & 1) * G->nranks,
                               G->cntseq_dev + (G->cnt_seq & 1), G->cnt_seq);
        } else {
            G->cnt_pending = false;
            HIP_TRY(hipMemcpyAsync(G->cnt_host, G->ex[0].gat_cnt, (size_t)G->nranks * sizeof(float), hipMemcpyDeviceToHost, G->stream));
        }
        if ((rc = exchange(XFER_GHOST_FULL)) != BLUB_OK) return rc;
        float mx = 0.0f; bool all_known = true;
        for (int k = 0; k < G->nranks; ++k) { mx = std::max(mx, G->cnt_host[k]); all_known = all_known && G->cnt_host[k] > 0.0f; }
        int np = SLAB_NP_DEFAULT;
        if (all_known) np = (int)(mx * 9.0f / 8.0f / (float)PCG_BPB) + 8;
        G->np_cur = std::max(128, std::min(SLAB_NP_MAX, (np + 7) & ~7));
    }
    if (RUNS(SS_TRANSFER)) {
        FOR_SLABS(stage_transfer(h, dt))
        if ((rc = slab_drop_ghosts(G)) != BLUB_OK) return rc;   // the velocity ghosts are only needed by the P2G gather
        if ((rc = slab_halo_velocity(G)) != BLUB_OK) return rc;
    }
    if (RUNS(SS_DIVERGENCE)) FOR_SLABS(stage_divergence(h))
    if (RUNS(SS_SOLVE_VELOCITY) && (rc = slab_solve(G, 0, dt)) != BLUB_OK) return rc;
    if (RUNS(SS_BINNING) && h0->rebin_freq != 0 && h0->step_counter % h0->rebin_freq == 0) FOR_SLABS(stage_binning(h))
    if (RUNS(SS_PROJECT)) {
        for (int i = 0; i < S; ++i) {
            blub_fluid* h = G->slabs[i];
            LAUNCH(h, KC_DIVERGENCE_REMOVE, k_divergence_remove_b, dim3(list_grids(h).active), dim3(BRICK_THREADS), h->bg, LIST(h, active), (const int8_t*)h->marker,
                   (const float*)h->pressure[0], (const float4*)h->solid, h->vel[0], h->vel[1], h->vel[2]);
        }
        if ((rc = slab_halo_velocity(G)) != BLUB_OK) return rc;
        FOR_SLABS(stage_extrapolate(h))
        if ((rc = slab_halo_velocity(G)) != BLUB_OK) return rc;
    }
    if (RUNS(SS_ADVECT)) FOR_SLABS(stage_advect_particles(h, dt, false))
    if (RUNS(SS_MIGRATE)) {
        if ((rc = exchange(XFER_MIGRATE)) != BLUB_OK) return rc;
        if ((rc = exchange(XFER_GHOST_POS)) != BLUB_OK) return rc;
        for (int i = 0; i < S; ++i) {   // marker + density list for own and ghost particles (advect_particles.comp:176-181)
            blub_fluid* h = G->slabs[i];
            const uint32_t n = h->num_particles + h->num_ghost;
            if (n) hipLaunchKernelGGL(k_slab_insert_density_ghosts, dim3(particle_blocks(n)), dim3(256), 0, G->stream, h->g, 0u, n, h->pos, h->marker, h->ll[0], (const uint32_t*)h->n_dev);
        }
        FOR_SLABS(build_lists_from_particles(h, COMPACT_STEP_B))
    }
    if (RUNS(SS_DENSITY_GATHER)) {
        FOR_SLABS(stage_density_gather(h, dt))
        if ((rc = slab_drop_ghosts(G)) != BLUB_OK) return rc;
    }
    if (RUNS(SS_SOLVE_DENSITY) && (rc = slab_solve(G, 1, dt)) != BLUB_OK) return rc;
    if (RUNS(SS_POSITION_CHANGE)) {
        for (int i = 0; i < S; ++i) {
            blub_fluid* h = G->slabs[i];
            LAUNCH(h, KC_POSITION_CHANGE, k_position_change_b, dim3(list_grids(h).active), dim3(BRICK_THREADS), h->bg, LIST(h, active), (const int8_t*)h->marker,
                   (const float*)h->pressure[1], dt, h->vel[0], h->vel[1], h->vel[2]);
        }
        if ((rc = slab_halo_velocity(G)) != BLUB_OK) return rc;
        FOR_SLABS(stage_extrapolate(h))
        if ((rc = slab_halo_velocity(G)) != BLUB_OK) return rc;
    }
    if (RUNS(SS_CORRECT)) FOR_SLABS(stage_correct(h))
    if (RUNS(SS_MIGRATE_B) && (rc = exchange(XFER_MIGRATE_B)) != BLUB_OK) return rc;
    if (RUNS(SS_FINISH)) for (int i = 0; i < S; ++i) { G->slabs[i]->step_counter += 1; (void)poll_stats(G->slabs[i], false, false); }
#undef RUNS
#undef FOR_SLABS
    return check_launch(h0);
}

static void slab_group_destroy(blub_slab_group* G) {
    if (!G) return;
    (void)hipSetDevice(G->device);
    if (G->stream) (void)hipStreamSynchronize(G->stream);
    auto F = [](void* p) { if (p) (void)hipFree(p); };
    for (void* m : G->ipc_opened) (void)hipIpcCloseMemHandle(m);
    for (void* a : G->ck_allocs) F(a);
    F(G->layer_hist);
    if (G->layer_hist_host) (void)hipHostFree(G->layer_hist_host);
    for (auto& ar : G->arena) F(ar.base);
    for (auto h : G->slabs) { h->n_dev = nullptr; destroy(h); }
    if (G->rec_host) (void)hipHostFree(G->rec_host);
    if (G->cntrec_host) (void)hipHostFree(G->cntrec_host);
    if (G->cntseq_host) (void)hipHostFree(G->cntseq_host);
    if (G->counts_host) (void)hipHostFree(G->counts_host);
    if (G->recv_host) (void)hipHostFree(G->recv_host);
    if (G->ctrl_host) (void)hipHostFree(G->ctrl_host);
    if (G->cnt_host) (void)hipHostFree(G->cnt_host);
    if (G->comm) (void)ncclCommDestroy(G->comm);   // (nullptr after an abort)
    if (G->stream) (void)hipStreamDestroy(G->stream);
    delete G;
}

// One generation of the restartable state of every local slab (see blub_slab.hip.h: SlabCheckpoint); generation = (step / interval) & 1
static int slab_checkpoint_alloc(blub_slab_group* G) {
    if (!G->ck[0].empty()) return BLUB_OK;
    for (int gen = 0; gen < 2; ++gen)
        for (auto h : G->slabs) {
            blubk::SlabCheckpoint c{};
            for (int k = 0; k < 4; ++k) { HIP_TRY(hipMalloc((void**)&c.part[k], (size_t)G->capacity * sizeof(float4))); G->ck_allocs.push_back(c.part[k]); }
            for (int k = 0; k < 2; ++k) { HIP_TRY(hipMalloc((void**)&c.pressure[k], h->vol_cells * sizeof(float))); G->ck_allocs.push_back(c.pressure[k]); }
            HIP_TRY(hipMalloc((void**)&c.n, 4 * sizeof(uint32_t))); G->ck_allocs.push_back(c.n);
            HIP_TRY(hipMalloc((void**)&c.step, sizeof(uint32_t))); G->ck_allocs.push_back(c.step);
            HIP_TRY(hipMemsetAsync(c.step, 0xFF, sizeof(uint32_t), G->stream));      // 0xFFFFFFFF: no generation yet
            HIP_TRY(hipMemsetAsync(c.n, 0, 4 * sizeof(uint32_t), G->stream));
            G->ck[gen].push_back(c);
        }
    return BLUB_OK;
}
static int slab_checkpoint(blub_slab_group* G) {
    int rc = slab_checkpoint_alloc(G);
    if (rc != BLUB_OK) return rc;
    const uint32_t step = G->slabs[0]->step_counter;
    const int gen = (int)((step / G->ck_interval) & 1u);
    for (size_t i = 0; i < G->slabs.size(); ++i) {
        blub_fluid* h = G->slabs[i];
        // (host-side counts are exact until the first asynchronous exchange replaced them by bounds; the device copy is current either way once uploaded)
        hipLaunchKernelGGL(blubk::k_slab_checkpoint, dim3(1024), dim3(256), 0, G->stream, G->ck[gen][i], (const float4*)h->pos, (const float4*)h->pvel[0], (const float4*)h->pvel[1],
                           (const float4*)h->pvel[2], (const float*)(h->pressure[0] + h->vol_first), (const float*)(h->pressure[1] + h->vol_first), (uint32_t)(h->vol_cells / 4),
                           (const uint32_t*)h->n_dev, h->num_particles, (const uint32_t*)G->dir_error[i], step);
    }
    return BLUB_OK;
}

// z-range of slab `index` of `nranks`: whole bricks, as even as possible
static void slab_range(int nz, int nranks, int index, int* z0, int* z1) {
    const int nbz = (nz + BZ - 1) / BZ;
    const int lo = (int)((int64_t)nbz * index / nranks), hi = (int)((int64_t)nbz * (index + 1) / nranks);
    *z0 = lo * BZ;
    *z1 = std::min(hi * BZ, index + 1 == nranks ? nz + BZ : hi * BZ);
}
// Cut planes of a group: the caller's (validated: cuts[0] = 0, strictly increasing multiples of the brick depth, the last one covers nz) or uniform
static int slab_cuts(int nz, int nranks, const int32_t* cuts, std::vector<int>& out) {
    const int nbz = (nz + BZ - 1) / BZ;
    out.assign((size_t)nranks + 1, 0);
    if (!cuts) { for (int r = 0; r < nranks; ++r) { int a, b; slab_range(nz, nranks, r, &a, &b); out[(size_t)r] = a; out[(size_t)r + 1] = b; } return BLUB_OK; }
    if (cuts[0] != 0) return set_error(BLUB_ERR_INVALID_ARGUMENT, "slab cuts: the first cut plane must be 0");
    for (int r = 0; r <= nranks; ++r) {
        if (cuts[r] % BZ != 0 && !(r == nranks && cuts[r] == nz)) return set_error(BLUB_ERR_INVALID_ARGUMENT, "slab cuts: every cut plane must be a multiple of the brick depth (4)");
        if (r > 0 && cuts[r] <= cuts[r - 1]) return set_error(BLUB_ERR_INVALID_ARGUMENT, "slab cuts: cut planes must increase strictly (every slab owns at least one brick layer)");
        out[(size_t)r] = cuts[r];
    }
    if (cuts[nranks] < nz || cuts[nranks] > nbz * BZ) return set_error(BLUB_ERR_INVALID_ARGUMENT, "slab cuts: the last cut plane must be the top of the grid");
    out[(size_t)nranks] = nbz * BZ;      // (whole bricks, like slab_range)
    return BLUB_OK;
}
// Contiguous partition of the brick layers into `nranks` slabs that minimises the heaviest slab (ties: the most even layer counts): dynamic programme
// over (slabs used, layers covered), O(nranks x layers^2) -- 64 .. 128 layers.  Every slab gets at least `min_layers`.
static int slab_partition_layers(const std::vector<double>& w, int nranks, int min_layers, std::vector<int>& first_layer) {
    const int L = (int)w.size();
    min_layers = std::max(1, min_layers);
    if ((int64_t)nranks * min_layers > L) return set_error(BLUB_ERR_INVALID_ARGUMENT, "more slabs (x minimum layers) than brick layers in z");
    std::vector<double> pre((size_t)L + 1, 0.0);
    for (int l = 0; l < L; ++l) pre[(size_t)l + 1] = pre[(size_t)l] + w[(size_t)l];
    struct Cost { double mx, sq; };
    auto better = [](const Cost& a, const Cost& b) { return a.mx < b.mx * (1.0 - 1e-12) || (a.mx <= b.mx * (1.0 + 1e-12) && a.sq < b.sq); };
    const Cost INF{1e300, 1e300};
    std::vector<std::vector<Cost>> best((size_t)nranks + 1, std::vector<Cost>((size_t)L + 1, INF));
    std::vector<std::vector<int>> from((size_t)nranks + 1, std::vector<int>((size_t)L + 1, -1));
    best[0][0] = Cost{0.0, 0.0};
    for (int k = 1; k <= nranks; ++k)
        for (int j = k * min_layers; j <= L - (nranks - k) * min_layers; ++j)
            for (int i = (k - 1) * min_layers; i <= j - min_layers; ++i) {
                if (best[(size_t)k - 1][(size_t)i].mx >= 1e300) continue;
                const double ww = pre[(size_t)j] - pre[(size_t)i];
                const Cost c{std::max(best[(size_t)k - 1][(size_t)i].mx, ww), best[(size_t)k - 1][(size_t)i].sq + (double)(j - i) * (double)(j - i)};
                if (better(c, best[(size_t)k][(size_t)j])) { best[(size_t)k][(size_t)j] = c; from[(size_t)k][(size_t)j] = i; }
            }
    first_layer.assign((size_t)nranks + 1, 0);
    int j = L;
    for (int k = nranks; k >= 1; --k) { first_layer[(size_t)k] = j; j = from[(size_t)k][(size_t)j]; if (j < 0) return set_error(BLUB_ERR_INVALID_ARGUMENT, "no partition found"); }
    return BLUB_OK;
}


static int slab_checkpoint(blub_slab_group* G);
// ---- moving the cut planes of a running group (round 5; review "missing" 6) -----------------------------------------------------------------------
// Precondition: BLUB_SLAB_FULL_VOLUMES (every slab holds every plane: nothing is reallocated, only ownership moves).  Every new cut lies strictly between
// its old neighbours, so state only moves between ADJACENT slabs: the planes of the two pressure volumes (warm starts) that change owner travel like halo
// planes, the particles through one ordinary migration exchange against the new ranges.  Everything else is scratch -- the brick
// bookkeeping is range-independent (k_bricks_build filters its work lists by the own range at every build).  Collective, between steps.
static int slab_recut(blub_slab_group* G, const std::vector<int>& nc) {
    const int S = (int)G->slabs.size();
    blub_fluid* h0 = G->slabs[0];
    const size_t pb = (size_t)h0->g.nx * h0->g.ny * sizeof(float);
    int rc = slab_refresh_counts(G);      // drains the stream; exact counts on the host (the synchronous exchange below takes them as exact)
    if (rc != BLUB_OK) return rc;
    // (1) pressure planes that change owner.  Cut r separates slab r - 1 (below) from slab r (above).
    auto moved = [&](int r, int* src, int* dst, int* za, int* zb) {
        const int o = G->cuts[(size_t)r], n = nc[(size_t)r];
        if (n == o) return false;
        if (n > o) { *src = r; *dst = r - 1; *za = o; *zb = n; } else { *src = r - 1; *dst = r; *za = n; *zb = o; }
        *zb = std::min(*zb, h0->g.nz);
        return *zb > *za;
    };
    G->comm_ops += 1;
    if (G->direct) G->flag_seq = seq_after(G->flag_seq);
    if (G->rccl && !G->direct) NCCL_TRY(ncclGroupStart());
    for (int r = 1; r < G->nranks; ++r) {
        int src, dst, za, zb;
        if (!moved(r, &src, &dst, &za, &zb)) continue;
        const size_t off = (size_t)za * pb, bytes = (size_t)(zb - za) * pb;
        const bool src_local = rank_local(G, src), dst_local = rank_local(G, dst);
        for (int w = 0; w < 2; ++w) {
            if (src_local) {
                const int i = src - G->first;
                char* mine = (char*)G->slabs[i]->pressure[w];
                if (G->direct) {
                    // (whole planes in pieces of < 4 GiB; the push kernel's copies are 32-bit sized)
                    for (size_t at = 0; at < bytes; at += (size_t)1 << 30) {
                        const size_t len = std::min(bytes - at, (size_t)1 << 30);
                        if ((rc = slab_copy(G, (char*)peer_vol(G, i, dst, mine, sizeof(float)) + off + at, mine + off + at, len)) != BLUB_OK) return rc;
                    }
                    push_flag_to(G, i, dst);
                } else if (dst_local) {
                    HIP_TRY(hipMemcpyAsync((char*)G->slabs[dst - G->first]->pressure[w] + off, mine + off, bytes, hipMemcpyDeviceToDevice, G->stream));
                } else NCCL_TRY(ncclSend(mine + off, bytes, ncclChar, dst, G->comm, G->stream));
            }
            if (dst_local && !src_local && !G->direct)      // (direct: the flag round below covers both neighbours of every slab)
                NCCL_TRY(ncclRecv((char*)G->slabs[dst - G->first]->pressure[w] + off, bytes, ncclChar, src, G->comm, G->stream));
        }
    }
    if (G->rccl && !G->direct) NCCL_TRY(ncclGroupEnd());
    if (G->direct) {
        // (every slab takes part in the flag round of this sequence number, whether or not it moves planes: the acknowledgement handshake is symmetric)
        for (int i = 0; i < S; ++i) { if (has_up(G, i)) push_flag_to(G, i, G->first + i + 1); if (has_down(G, i)) push_flag_to(G, i, G->first + i - 1); }
        if ((rc = slab_copy_flush(G)) != BLUB_OK) return rc;
        for (int i = 0; i < S; ++i) if ((rc = slab_wait(G, i, neighbour_mask(G, i))) != BLUB_OK) return rc;
    }
    // (2) the new ranges, then ONE migration exchange against them (synchronous variant: no message-size history applies to a re-cut)
    G->cuts = nc;
    for (int i = 0; i < S; ++i) {
        blub_fluid* h = G->slabs[i];
        h->slab_z0 = nc[(size_t)G->first + i]; h->slab_z1 = nc[(size_t)G->first + i + 1];
        h->have_last_counts = false;
    }
    // the ghost planes of the two pressure volumes at the NEW interfaces (the next solves start from p with its halo: pressure_init.comp:50-83)
    for (int w = 0; w < 2; ++w)
        if ((rc = slab_halo(G, {[w](blub_fluid* h) { return (void*)h->pressure[w]; }}, 4)) != BLUB_OK) return rc;
    const bool was_direct = G->direct;
    G->direct = false;      // (the synchronous exchange speaks device copies / RCCL; a multi-process direct group still holds its communicator)
    rc = slab_exchange_particles(G, XFER_MIGRATE);
    G->direct = was_direct;
    if (rc != BLUB_OK) return rc;
    for (auto& H : G->hist) H = blub_slab_group::Hist();
    G->cnt_pending = false;
    for (int k = 0; k < G->nranks; ++k) G->cnt_host[k] = 0.0f;      // (the gathered brick counts belong to the old ranges)
    // checkpoint generations taken under the OLD ranges would put particles and pressure planes back where they no longer belong: drop them and, if
    // checkpoints are in use, take a fresh generation of the re-cut state right away
    if (!G->ck[0].empty()) {
        for (int gen = 0; gen < 2; ++gen) for (auto& c : G->ck[gen]) HIP_TRY(hipMemsetAsync(c.step, 0xFF, sizeof(uint32_t), G->stream));
        if (G->ck_interval && (rc = slab_checkpoint(G)) != BLUB_OK) return rc;
    }
    HIP_TRY(hipStreamSynchronize(G->stream));
    return BLUB_OK;
}
// FLUID bricks per brick layer of the whole domain, the same array on every rank: every local slab counts its own layers, the segments are gathered
static int slab_layer_histogram(blub_slab_group* G, std::vector<double>& hist) {
    blub_fluid* h0 = G->slabs[0];
    const int L = h0->bg.nbz, S = (int)G->slabs.size();
    if (!G->layer_hist) {
        HIP_TRY(hipMalloc((void**)&G->layer_hist, (size_t)S * G->nranks * L * sizeof(float)));
        HIP_TRY(hipHostMalloc((void**)&G->layer_hist_host, (size_t)G->nranks * L * sizeof(float)));
    }
    // one array of nranks x L per local slab (segment r = rank r's counts), gathered with the group's own transport
    HIP_TRY(hipMemsetAsync(G->layer_hist, 0, (size_t)S * G->nranks * L * sizeof(float), G->stream));
    for (int i = 0; i < S; ++i) {
        blub_fluid* h = G->slabs[i];
        float* mine = G->layer_hist + ((size_t)i * G->nranks + (size_t)(G->first + i)) * L;
        hipLaunchKernelGGL(blubk::k_slab_layer_histogram, dim3(64), dim3(256), 0, G->stream, h->bg, (const uint32_t*)h->list_fluid, (const uint32_t*)&h->counts->n_fluid, mine);
    }
    const bool was_direct = G->direct;
    if (G->direct && (int)G->slabs.size() != G->nranks) G->direct = false;      // (between processes the histogram is not inside an exportable region: RCCL carries it)
    float* base = G->layer_hist;
    const int nr = G->nranks;
    int rc = G->direct ? BLUB_OK : slab_gather(G, [base, nr, L](int i) { return base + (size_t)i * nr * L; }, L);
    if (G->direct) {      // local group: plain copies between the local arrays
        for (int sidx = 0; sidx < S && rc == BLUB_OK; ++sidx)
            for (int d = 0; d < S && rc == BLUB_OK; ++d)
                if (d != sidx) HIP_TRY(hipMemcpyAsync(base + ((size_t)d * nr + sidx) * L, base + ((size_t)sidx * nr + sidx) * L, (size_t)L * sizeof(float), hipMemcpyDeviceToDevice, G->stream));
    }
    G->direct = was_direct;
    if (rc != BLUB_OK) return rc;
    HIP_TRY(hipMemcpyAsync(G->layer_hist_host, G->layer_hist, (size_t)nr * L * sizeof(float), hipMemcpyDeviceToHost, G->stream));
    HIP_TRY(hipStreamSynchronize(G->stream));
    hist.assign((size_t)L, 0.0);
    for (int r = 0; r < nr; ++r) for (int l = 0; l < L; ++l) hist[(size_t)l] += G->layer_hist_host[(size_t)r * L + l];
    return BLUB_OK;
}

// RCCL transport of the PCG partials, chosen by measurement on the hardware at hand (the two candidates cannot be ranked
// on the 1-GPU development box): per candidate, 30 rounds of what one PCG iteration issues; the slowest rank's time decides
// (all-reduced, so every rank picks the same mode).  blub_slab_group_set_gather_mode overrides.
static int slab_calibrate(blub_slab_group* G) {
    snprintf(G->transport, sizeof G->transport, "rccl, %d ranks", G->nranks);
    if (G->nranks == 1) return BLUB_OK;
    blub_fluid* h = G->slabs[0];
    auto& e = G->ex[0];
    const int np = SLAB_NP_DEFAULT;
    hipEvent_t a, b;
    HIP_TRY(hipEventCreate(&a)); HIP_TRY(hipEventCreate(&b));
    float* times = nullptr;
    HIP_TRY(hipMalloc((void**)&times, 2 * sizeof(float)));
    float ms[2] = {0.f, 0.f};
    int rc = BLUB_OK;
    auto round = [&]() -> int {
        int r2 = slab_gather(G, [&](int) { return e.gat_dir; }, np);
        if (r2 != BLUB_OK) return r2;
        return slab_fused(G, [&]() -> int { return slab_halo(G, {[](blub_fluid* f) { return (void*)f->residual; }}, 4, false); },
                          [&](bool own) { return slab_gather(G, [&](int) { return reinterpret_cast<float*>(e.gat_upd); }, 2 * np, own); });
    };
    for (int mode = 0; mode < 2 && rc == BLUB_OK; ++mode) {
        G->gather_mode = mode;
        for (int k = 0; k < 5 && rc == BLUB_OK; ++k) rc = round();
        if (rc != BLUB_OK) break;
        HIP_TRY(hipEventRecord(a, G->stream));
        for (int k = 0; k < 30 && rc == BLUB_OK; ++k) rc = round();
        HIP_TRY(hipEventRecord(b, G->stream));
        HIP_TRY(hipEventSynchronize(b));
        HIP_TRY(hipEventElapsedTime(&ms[mode], a, b));
    }
    if (rc == BLUB_OK) {
        HIP_TRY(hipMemcpyAsync(times, ms, sizeof ms, hipMemcpyHostToDevice, G->stream));
        NCCL_TRY(ncclAllReduce(times, times, 2, ncclFloat, ncclMax, G->comm, G->stream));
        HIP_TRY(hipMemcpyAsync(ms, times, sizeof ms, hipMemcpyDeviceToHost, G->stream));
        HIP_TRY(hipStreamSynchronize(G->stream));
        G->gather_mode = ms[1] < ms[0] ? 1 : 0;
        snprintf(G->transport, sizeof G->transport, "rccl, %d ranks, partials by %s (calibrated: grouped send/recv %.1f us, ncclAllGather %.1f us per PCG iteration)",
                 G->nranks, G->gather_mode ? "ncclAllGather" : "grouped send/recv", ms[0] / 30.f * 1e3f, ms[1] / 30.f * 1e3f);
    }
    (void)hipFree(times); (void)hipEventDestroy(a); (void)hipEventDestroy(b);
    (void)h;
    G->comm_ops = 0;
    // the calibration traffic went through the residual ghost planes and the gather arrays: put them back to zero
    { int rz = vol_zero(G->slabs[0], G->slabs[0]->residual); if (rz != BLUB_OK) return rz; }
    HIP_TRY(hipMemsetAsync(e.gat_dir, 0, (size_t)G->nranks * np * sizeof(float), G->stream));
    HIP_TRY(hipMemsetAsync(e.gat_upd, 0, (size_t)G->nranks * np * sizeof(float2), G->stream));
    HIP_TRY(hipStreamSynchronize(G->stream));
    return rc;
}

static int slab_group_create(const blub_fluid_desc* d, int nranks, int first, int nlocal, const void* nccl_id, blub_slab_group** out, const int32_t* cuts = nullptr, uint32_t mem_mode = 0) {
    if (!d || !out || nranks < 1 || nlocal < 1 || first < 0 || first + nlocal > nranks || nlocal > 8) return set_error(BLUB_ERR_INVALID_ARGUMENT, "bad slab group arguments");
    *out = nullptr;
    if ((int)((d->nz + BZ - 1) / BZ) < nranks) return set_error(BLUB_ERR_INVALID_ARGUMENT, "more slabs than brick layers in z");
    const bool full_volumes = (mem_mode & BLUB_SLAB_FULL_VOLUMES) != 0;      // every slab holds every plane: the cut planes may move later (blub_slab_group_recut)
    mem_mode &= ~(uint32_t)BLUB_SLAB_FULL_VOLUMES;
    if (mem_mode > BLUB_SLAB_MEMORY_UNCACHED) return set_error(BLUB_ERR_INVALID_ARGUMENT, "bad slab memory mode");
    std::vector<int> cut_planes;
    { int rcc = slab_cuts((int)d->nz, nranks, cuts, cut_planes); if (rcc != BLUB_OK) return rcc; }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) return set_error(BLUB_ERR_NO_DEVICE, "no HIP device (libblubhip has no CPU fallback)");
    int dev = d->device;
    if (dev < 0) HIP_TRY(hipGetDevice(&dev));
    HIP_TRY(hipSetDevice(dev));
    blub_slab_group* G = new (std::nothrow) blub_slab_group();
    if (!G) return set_error(BLUB_ERR_OUT_OF_MEMORY, "host allocation failed");
    G->nranks = nranks; G->first = first; G->device = dev; G->capacity = std::max<uint32_t>(d->max_num_particles, 1);
    G->rccl = nccl_id != nullptr;
    G->cuts = cut_planes; G->mem_mode = (int)mem_mode; G->full_volumes = full_volumes || nranks == 1;
    int rc = BLUB_OK;
    if (hipStreamCreateWithFlags(&G->stream, hipStreamNonBlocking) != hipSuccess) { delete G; return set_error(BLUB_ERR_DEVICE, "hipStreamCreate failed"); }
    blub_fluid_desc dd = *d; dd.device = dev;
    // Slab-local volumes: a slab holds its own planes plus two brick layers on either side -- ghost particles (GHOST_MARGIN cells beyond the interface)
    // mark the neighbouring brick layer FLUID, its dilation makes the layer behind that one ACTIVE, and the reset kernels clear every brick that
    // ever was -- and the SAME number of planes on every rank so that the exportable regions have one layout.
    int vol_planes = 0;
    G->vol_z0_of.assign((size_t)nranks, 0); G->vol_first_of.assign((size_t)nranks, 0);
    for (int r = 0; r < nranks; ++r) {
        const int a = G->cuts[(size_t)r], b = G->cuts[(size_t)r + 1];
        const int za = std::max(0, a - 2 * BZ), zb = std::min((int)d->nz, std::min(b, (int)d->nz) + 2 * BZ);
        G->vol_z0_of[(size_t)r] = za; G->vol_first_of[(size_t)r] = (size_t)d->nx * d->ny * (size_t)za;
        vol_planes = std::max(vol_planes, zb - za);
    }
    if (nranks == 1 || full_volumes) {    // (whole grid: 8.5 GiB per rank at 512^3 -- of 288)
        vol_planes = 0;
        std::fill(G->vol_z0_of.begin(), G->vol_z0_of.end(), 0); std::fill(G->vol_first_of.begin(), G->vol_first_of.end(), (size_t)0);
    }
    for (int i = 0; i < nlocal && rc == BLUB_OK; ++i) {
        blub_fluid* h = nullptr;
        rc = create(&dd, &h, G->stream, G->vol_z0_of[(size_t)first + i], vol_planes, (int)mem_mode);
        if (rc != BLUB_OK) break;
        h->slab_z0 = G->cuts[(size_t)first + i]; h->slab_z1 = G->cuts[(size_t)first + i + 1];
        h->max_steps_in_flight = 0;   // every particle exchange synchronises the host anyway
        G->slabs.push_back(h);
        blub_slab_group::Extra e;
        const size_t P = G->capacity;
        // everything a particle exchange or a gather touches lives in ONE allocation per slab (the "arena"): with the direct transport the
        // neighbours write into it, so it is one of the slab's exportable regions (one hipIpc handle)
        blub_slab_group::Arena ar;
        ar.bytes = (size_t)3 * P * 4 + (size_t)4 * (P + 1) * 16 + (size_t)12 * P * 16 + (size_t)nranks * blubk::SLAB_NP_MAX * (4 + 8 + 16 + 16) + (size_t)nranks * 8 + 64 * 1024;
        if (rc == BLUB_OK && shared_malloc((void**)&ar.base, ar.bytes, (int)mem_mode) != hipSuccess) rc = set_error(BLUB_ERR_OUT_OF_MEMORY, "allocation of a slab's exchange arena failed");
        if (rc == BLUB_OK && hipMemsetAsync(ar.base, 0, ar.bytes, G->stream) != hipSuccess) rc = set_error(BLUB_ERR_DEVICE, "hipMemsetAsync failed");
        auto sub = [&](auto** pp, size_t count) {
            using T = std::remove_pointer_t<std::remove_pointer_t<decltype(pp)>>;
            if (rc != BLUB_OK) return;
            const size_t at = (ar.used + 255) & ~(size_t)255;
            if (at + count * sizeof(T) > ar.bytes) { rc = set_error(BLUB_ERR_OUT_OF_MEMORY, "slab exchange arena exhausted"); return; }
            *pp = reinterpret_cast<T*>(ar.base + at);
            ar.used = at + count * sizeof(T);
        };
        sub(&e.leave_idx, P); sub(&e.hole_idx, P); sub(&e.fill_idx, P);
        // send buffers (position messages carry one float4 of header in front) and the staging of what arrives
        sub(&e.up_msg, P + 1); sub(&e.dn_msg, P + 1); sub(&e.rb_msg, P + 1); sub(&e.ra_msg, P + 1);
        if (rc == BLUB_OK) { e.up[0] = e.up_msg + 1; e.dn[0] = e.dn_msg + 1; e.rb[0] = e.rb_msg + 1; e.ra[0] = e.ra_msg + 1; }
        for (int k = 1; k < 4; ++k) { sub(&e.up[k], P); sub(&e.dn[k], P); sub(&e.rb[k], P); sub(&e.ra[k], P); }
        sub(&e.append_done, 1);
        sub(&h->n_dev, 4);
        sub(&e.counts, 1); sub(&e.recv_counts, 2);
        sub(&e.gat_dir, (size_t)nranks * blubk::SLAB_NP_MAX); sub(&e.gat_upd, (size_t)nranks * blubk::SLAB_NP_MAX);
        sub(&e.gat_cnt, (size_t)nranks);
        sub(&e.gat4[0], (size_t)nranks * blubk::SLAB_NP_MAX); sub(&e.gat4[1], (size_t)nranks * blubk::SLAB_NP_MAX);
        uint32_t *fl = nullptr, *bd = nullptr, *de = nullptr;
        sub(&fl, 64); sub(&bd, 16); sub(&de, 16);
        G->flags.push_back(fl); G->blocks_done.push_back(bd); G->dir_error.push_back(de);
        G->arena.push_back(ar);
        if (rc == BLUB_OK) rc = ensure_pcg1_buffers(h);      // (eager: the direct transport exports them)
        // exportable regions, the same list on every rank: the volume slab, the three extra PCG volumes, the arena
        std::vector<blub_slab_group::Region> regs;
        if (rc == BLUB_OK) {
            regs.push_back({h->slab, h->slab_bytes});
            for (int k = 0; k < 3; ++k) regs.push_back({reinterpret_cast<char*>(h->cgbuf_alloc[k]), h->vol_cells * sizeof(float)});
            regs.push_back({ar.base, ar.bytes});
        }
        G->regions.push_back(regs);
        G->ex.push_back(e);
    }
    // peer view: the regions of every rank as addressable from here -- local slabs by their own pointers; remote ones after blub_slab_group_connect
    if (rc == BLUB_OK) {
        G->peer_base.assign((size_t)nranks, std::vector<char*>());
        for (int i = 0; i < (int)G->slabs.size(); ++i) for (auto& rg : G->regions[i]) G->peer_base[(size_t)first + i].push_back(rg.base);
    }
    if (rc == BLUB_OK && hipHostMalloc((void**)&G->counts_host, nlocal * sizeof(blubk::SlabCounts)) != hipSuccess) rc = set_error(BLUB_ERR_OUT_OF_MEMORY, "hipHostMalloc failed");
    if (rc == BLUB_OK && hipHostMalloc((void**)&G->recv_host, 2 * nlocal * sizeof(uint32_t)) != hipSuccess) rc = set_error(BLUB_ERR_OUT_OF_MEMORY, "hipHostMalloc failed");
    if (rc == BLUB_OK && hipHostMalloc((void**)&G->ctrl_host, 2 * sizeof(blubk::PcgCtrl)) != hipSuccess) rc = set_error(BLUB_ERR_OUT_OF_MEMORY, "hipHostMalloc failed");
    if (rc == BLUB_OK && hipHostMalloc((void**)&G->cnt_host, (size_t)(nranks + nlocal) * sizeof(float)) != hipSuccess) rc = set_error(BLUB_ERR_OUT_OF_MEMORY, "hipHostMalloc failed");
    if (rc == BLUB_OK) memset(G->cnt_host, 0, (size_t)(nranks + nlocal) * sizeof(float));
    G->hist.assign((size_t)nlocal * XFER_KINDS, blub_slab_group::Hist());
    const size_t nrec = (size_t)nlocal * XFER_KINDS * XFER_RING;
    if (rc == BLUB_OK && (hipHostMalloc((void**)&G->rec_host, nrec * sizeof(blubk::SlabXferRecord), hipHostMallocMapped) != hipSuccess ||
                          hipHostGetDevicePointer((void**)&G->rec_dev, G->rec_host, 0) != hipSuccess)) rc = set_error(BLUB_ERR_OUT_OF_MEMORY, "hipHostMalloc failed");
    if (rc == BLUB_OK) memset(G->rec_host, 0, nrec * sizeof(blubk::SlabXferRecord));
    if (rc == BLUB_OK && (hipHostMalloc((void**)&G->cntrec_host, 2 * (size_t)nranks * sizeof(float), hipHostMallocMapped) != hipSuccess ||
                          hipHostGetDevicePointer((void**)&G->cntrec_dev, G->cntrec_host, 0) != hipSuccess)) rc = set_error(BLUB_ERR_OUT_OF_MEMORY, "hipHostMalloc failed");
    if (rc == BLUB_OK && (hipHostMalloc((void**)&G->cntseq_host, 2 * sizeof(uint32_t), hipHostMallocMapped) != hipSuccess ||
                          hipHostGetDevicePointer((void**)&G->cntseq_dev, G->cntseq_host, 0) != hipSuccess)) rc = set_error(BLUB_ERR_OUT_OF_MEMORY, "hipHostMalloc failed");
    if (rc == BLUB_OK) { memset(G->cntrec_host, 0, 2 * (size_t)nranks * sizeof(float)); memset(G->cntseq_host, 0, 2 * sizeof(uint32_t)); }
    if (rc == BLUB_OK) { memset(G->counts_host, 0, nlocal * sizeof(blubk::SlabCounts)); memset(G->recv_host, 0, 2 * nlocal * sizeof(uint32_t)); memset(G->ctrl_host, 0, 2 * sizeof(blubk::PcgCtrl)); }
    if (rc == BLUB_OK && G->rccl) {
        if (nlocal != 1) rc = set_error(BLUB_ERR_INVALID_ARGUMENT, "RCCL slab groups hold exactly one slab per process");
        else {
            ncclUniqueId id; memcpy(&id, nccl_id, sizeof(id));
            ncclResult_t r = ncclCommInitRank(&G->comm, nranks, id, first);
            if (r != ncclSuccess) rc = set_error(BLUB_ERR_COMM, ncclGetErrorString(r));
        }
    }
    if (rc == BLUB_OK && hipStreamSynchronize(G->stream) != hipSuccess) rc = set_error(BLUB_ERR_DEVICE, "slab group initialisation failed");
    if (rc == BLUB_OK && G->rccl) rc = slab_calibrate(G);
    if (rc != BLUB_OK) { std::string keep = g_last_error; slab_group_destroy(G); g_last_error = keep; return rc; }
    *out = G;
    return BLUB_OK;
}

}  // namespace blub

extern "C" {
int blub_slab_group_set_transport(blub_slab_group* g, int kind);
int blub_rccl_unique_id(void* out128) {
    if (!out128) return blub::set_error(BLUB_ERR_INVALID_ARGUMENT, "null argument");
    static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId is expected to be 128 bytes");
    ncclUniqueId id;
    ncclResult_t r = ncclGetUniqueId(&id);
    if (r != ncclSuccess) return blub::set_error(BLUB_ERR_COMM, ncclGetErrorString(r));
    memcpy(out128, &id, sizeof(id));
    return BLUB_OK;
}
int blub_slab_range(uint32_t nz, int num_slabs, int index, int32_t* z0, int32_t* z1) {   // host only
    if (!z0 || !z1 || num_slabs < 1 || index < 0 || index >= num_slabs) return blub::set_error(BLUB_ERR_INVALID_ARGUMENT, "bad argument");
    int a, b; blub::slab_range((int)nz, num_slabs, index, &a, &b);
    *z0 = a; *z1 = std::min<int>(b, (int)nz);
    return BLUB_OK;
}
int blub_slab_group_create_local(const blub_fluid_desc* desc, int num_slabs, blub_slab_group** out) { return blub_slab_group_create_local_ex(desc, num_slabs, nullptr, BLUB_SLAB_MEMORY_COARSE, out); }
int blub_slab_group_create_local_ex(const blub_fluid_desc* desc, int num_slabs, const int32_t* cuts, uint32_t memory_mode, blub_slab_group** out) {
    int rc = blub::slab_group_create(desc, num_slabs, 0, num_slabs, nullptr, out, cuts, memory_mode);
    if (rc != BLUB_OK) return rc;
    // The direct transport is the default where it is available (every slab's volumes in one allocation -- a slab that fell back to per-volume
    // allocations under memory pressure has none --, at most 8 slabs); otherwise the group keeps the host-issued copies it was created with.  A
    // creation that succeeded never turns into an error here (round-4 ADVICE: it used to return the status of set_transport with *out still set).
    bool can = num_slabs - 1 <= blubk::SLAB_MAX_PEERS;
    for (auto h : (*out)->slabs) can = can && h->slab != nullptr;
    if (can && blub_slab_group_set_transport(*out, 1) != BLUB_OK) (void)blub_slab_group_set_transport(*out, 0);
    return BLUB_OK;
}
int blub_slab_group_create_rccl(const blub_fluid_desc* desc, int rank, int num_ranks, const void* unique_id_128, blub_slab_group** out) {
    return blub_slab_group_create_rccl_ex(desc, rank, num_ranks, unique_id_128, nullptr, BLUB_SLAB_MEMORY_COARSE, out);
}
int blub_slab_group_create_rccl_ex(const blub_fluid_desc* desc, int rank, int num_ranks, const void* unique_id_128, const int32_t* cuts, uint32_t memory_mode, blub_slab_group** out) {
    if (!unique_id_128) return blub::set_error(BLUB_ERR_INVALID_ARGUMENT, "null unique id");
    return blub::slab_group_create(desc, num_ranks, rank, 1, unique_id_128, out, cuts, memory_mode);
}
// Host only.  Cut planes that give every slab about the same number of FLUID bricks (the unit the PCG kernels, the list walks and -- through the
// particles per brick -- the particle kernels scale with): weight of a brick layer = distinct bricks of it that hold a particle, plus a small constant so
// that empty stretches are shared out instead of all going to one neighbour.  The headline scene's two dams sit in z < 32 and z >= 224 of 256: uniform
// cuts into 8 leave six slabs without fluid.
int blub_slab_balanced_cuts(const uint32_t grid_dim[3], uint32_t n, const float* pos_ll, int num_slabs, int min_layers, int32_t* cuts_out, uint32_t* fluid_bricks_out) {
    if (!grid_dim || (n && !pos_ll) || !cuts_out || num_slabs < 1) return blub::set_error(BLUB_ERR_INVALID_ARGUMENT, "bad argument");
    const int nx = (int)grid_dim[0], ny = (int)grid_dim[1], nz = (int)grid_dim[2];
    const int nbx = (nx + blubk::BX - 1) / blubk::BX, nby = (ny + blubk::BY - 1) / blubk::BY, nbz = (nz + blubk::BZ - 1) / blubk::BZ;
    if (nbz < num_slabs) return blub::set_error(BLUB_ERR_INVALID_ARGUMENT, "more slabs than brick layers in z");
    std::vector<uint8_t> mark((size_t)nbx * nby * nbz, 0);
    for (uint32_t i = 0; i < n; ++i) {
        const float fx = pos_ll[4 * (size_t)i], fy = pos_ll[4 * (size_t)i + 1], fz = pos_ll[4 * (size_t)i + 2];
        if (!(fx >= 0.0f && fy >= 0.0f && fz >= 0.0f && fx < (float)nx && fy < (float)ny && fz < (float)nz)) continue;      // (also: NaN)
        const int x = (int)fx, y = (int)fy, z = (int)fz;
        mark[((size_t)(z / blubk::BZ) * nby + (size_t)(y / blubk::BY)) * nbx + (size_t)(x / blubk::BX)] = 1;
    }
    std::vector<double> w((size_t)nbz, 0.0);
    double total = 0.0;
    for (int l = 0; l < nbz; ++l) { for (size_t k = 0; k < (size_t)nbx * nby; ++k) w[(size_t)l] += mark[(size_t)l * nbx * nby + k]; total += w[(size_t)l]; }
    std::vector<double> bricks = w;
    const double eps = std::max(1.0, total) / (double)nbz * 0.02;
    for (auto& v : w) v += eps;
    std::vector<int> first;
    int rc = blub::slab_partition_layers(w, num_slabs, min_layers, first);
    if (rc != BLUB_OK) return rc;
    for (int r = 0; r <= num_slabs; ++r) cuts_out[r] = first[(size_t)r] * blubk::BZ;
    if (fluid_bricks_out)
        for (int r = 0; r < num_slabs; ++r) { double b = 0.0; for (int l = first[(size_t)r]; l < first[(size_t)r + 1]; ++l) b += bricks[(size_t)l]; fluid_bricks_out[r] = (uint32_t)b; }
    return BLUB_OK;
}
// ---- checkpoints and in-place recovery (direct transport) ----------------------------------------------------------------------------------------
int blub_slab_group_set_checkpoint_interval(blub_slab_group* g, uint32_t every_n_steps) {
    if (!g) return blub::set_error(BLUB_ERR_INVALID_ARGUMENT, "null handle");
    if (hipSetDevice(g->device) != hipSuccess) return blub::set_error(BLUB_ERR_DEVICE, "hipSetDevice failed");
    g->ck_interval = every_n_steps;
    return every_n_steps ? blub::slab_checkpoint_alloc(g) : BLUB_OK;
}
int blub_slab_group_checkpoints(blub_slab_group* g, uint32_t steps_out[2]) {
    if (!g || !steps_out) return blub::set_error(BLUB_ERR_INVALID_ARGUMENT, "null argument");
    steps_out[0] = steps_out[1] = 0xFFFFFFFFu;
    if (g->ck[0].empty()) return BLUB_OK;
    if (hipSetDevice(g->device) != hipSuccess) return blub::set_error(BLUB_ERR_DEVICE, "hipSetDevice failed");
    HIP_TRY(hipStreamSynchronize(g->stream));
    for (int gen = 0; gen < 2; ++gen) {
        // a generation counts only if EVERY local slab holds the same step (they are written together; a slab whose time-out mark was set skipped it)
        uint32_t common = 0xFFFFFFFFu; bool first = true;
        for (auto& c : g->ck[gen]) {
            uint32_t v = 0xFFFFFFFFu;
            HIP_TRY(hipMemcpy(&v, c.step, sizeof v, hipMemcpyDeviceToHost));
            if (first) { common = v; first = false; } else if (v != common) common = 0xFFFFFFFFu;
        }
        steps_out[gen] = common;
    }
    return BLUB_OK;
}
int blub_slab_group_exchange_sequence(const blub_slab_group* g, uint32_t* seq_out) {
    if (!g || !seq_out) return blub::set_error(BLUB_ERR_INVALID_ARGUMENT, "null argument");
    *seq_out = g->flag_seq;
    return BLUB_OK;
}
// Collective (every rank, the same arguments), after EVERY rank has drained its stream (blub_slab_group_synchronize -- which reports and clears the
// time-out -- then a barrier of the caller's control plane): back to the checkpoint of `step`, exchange sequence numbers restarted at `sequence_base`
// (larger than any number a rank has issued: the ranks stopped at different points of the step that failed).
int blub_slab_group_restore(blub_slab_group* g, uint32_t step, uint32_t sequence_base) {
    if (!g) return blub::set_error(BLUB_ERR_INVALID_ARGUMENT, "null handle");
    if (hipSetDevice(g->device) != hipSuccess) return blub::set_error(BLUB_ERR_DEVICE, "hipSetDevice failed");
    if (g->rccl && !g->direct) return blub::set_error(BLUB_ERR_UNSUPPORTED, "in-place recovery is for the direct transport (a failed RCCL operation aborts the communicator)");
    uint32_t have[2];
    { int rc = blub_slab_group_checkpoints(g, have); if (rc != BLUB_OK) return rc; }
    int gen = -1;
    for (int k = 0; k < 2; ++k) if (have[k] == step && step != 0xFFFFFFFFu) gen = k;
    if (gen < 0) return blub::set_error(BLUB_ERR_INVALID_ARGUMENT, "no checkpoint of that step is held (blub_slab_group_checkpoints)");
    for (size_t i = 0; i < g->slabs.size(); ++i) {
        blub_fluid* h = g->slabs[i];
        HIP_TRY(hipMemsetAsync(g->dir_error[i], 0, sizeof(uint32_t), g->stream));
        hipLaunchKernelGGL(blubk::k_slab_restore, dim3(1024), dim3(256), 0, g->stream, g->ck[gen][i], h->pos, h->pvel[0], h->pvel[1], h->pvel[2],
                           h->pressure[0] + h->vol_first, h->pressure[1] + h->vol_first, (uint32_t)(h->vol_cells / 4), h->n_dev);
        h->step_counter = step; h->num_ghost = 0; h->all_touched = true; h->bricks_premarked = false; h->have_last_counts = false;
        h->pressure_initialised[0] = h->pressure_initialised[1] = true;      // (the checkpoint holds them as they were: zero if no solve had run)
        HIP_TRY(hipMemsetAsync(h->brick_fluid, 0, (size_t)h->bg.nb, g->stream));
        h->stats_pending[0].clear(); h->stats_pending[1].clear(); h->stats_dt[0].clear(); h->stats_dt[1].clear();      // samples of the abandoned steps never land
    }
    for (auto& H : g->hist) H = blub_slab_group::Hist();
    g->cnt_pending = false; g->ctrl_host_valid[0] = g->ctrl_host_valid[1] = false;
    g->copies.n = 0; g->push_flags.n = 0; g->push_flags.n_ack = 0;
    g->flag_seq = sequence_base ? sequence_base : 1u;
    return blub::slab_refresh_counts(g);      // (drains the stream; exact counts back on the host)
}
int blub_slab_group_recut(blub_slab_group* g, const int32_t* new_cuts) {
    if (!g || !new_cuts) return blub::set_error(BLUB_ERR_INVALID_ARGUMENT, "null argument");
    if (hipSetDevice(g->device) != hipSuccess) return blub::set_error(BLUB_ERR_DEVICE, "hipSetDevice failed");
    if (!g->full_volumes) return blub::set_error(BLUB_ERR_UNSUPPORTED, "moving the cut planes needs a group created with BLUB_SLAB_FULL_VOLUMES");
    if (g->rccl && !g->comm) return blub::set_error(BLUB_ERR_COMM, "the group's communicator was aborted after an earlier failure");
    // a multi-process direct group: what follows are UNBOUNDED collective operations -- a rank whose peer has already left the step loop after a time-out
    // must not enter them (round-5 ADVICE): drain the stream and report a pending time-out first (BLUB_ERR_COMM: recoverable in place, like in a step)
    if (g->direct && g->rccl) { int rc0 = blub_slab_group_synchronize(g); if (rc0 != BLUB_OK) return rc0; }
    std::vector<int> nc;
    { int rc = blub::slab_cuts(g->slabs[0]->g.nz, g->nranks, new_cuts, nc); if (rc != BLUB_OK) return rc; }
    for (int r = 1; r < g->nranks; ++r)
        if (nc[(size_t)r] <= g->cuts[(size_t)r - 1] || nc[(size_t)r] >= g->cuts[(size_t)r + 1])
            return blub::set_error(BLUB_ERR_INVALID_ARGUMENT, "a cut plane may only move strictly between its old neighbours (state moves between adjacent slabs only): re-cut in several steps");
    if (nc == g->cuts) return BLUB_OK;
    return blub::slab_recut(g, nc);
}
// Collective.  FLUID bricks per brick layer are counted on the device and gathered; every rank derives the same balanced cuts (the partition of
// blub_slab_balanced_cuts), limited to what one re-cut may move, and the group re-cuts when that lowers the heaviest slab's share by more than 5 %.
// *changed (may be NULL): 1 if the cuts moved.  One host synchronisation: call it every few dozen steps, not every step.
int blub_slab_group_rebalance(blub_slab_group* g, int min_layers, int* changed) {
    if (!g) return blub::set_error(BLUB_ERR_INVALID_ARGUMENT, "null handle");
    if (changed) *changed = 0;
    if (hipSetDevice(g->device) != hipSuccess) return blub::set_error(BLUB_ERR_DEVICE, "hipSetDevice failed");
    if (!g->full_volumes) return blub::set_error(BLUB_ERR_UNSUPPORTED, "moving the cut planes needs a group created with BLUB_SLAB_FULL_VOLUMES");
    if (g->nranks == 1) return BLUB_OK;
    if (g->direct && g->rccl) { int rc0 = blub_slab_group_synchronize(g); if (rc0 != BLUB_OK) return rc0; }      // (see blub_slab_group_recut)
    std::vector<double> bricks;
    { int rc = blub::slab_layer_histogram(g, bricks); if (rc != BLUB_OK) return rc; }
    const int L = (int)bricks.size();
    double total = 0.0; for (double v : bricks) total += v;
    std::vector<double> w = bricks;
    const double eps = std::max(1.0, total) / (double)L * 0.02;
    for (auto& v : w) v += eps;
    std::vector<int> first;
    { int rc = blub::slab_partition_layers(w, g->nranks, min_layers, first); if (rc != BLUB_OK) return rc; }
    std::vector<int32_t> nc((size_t)g->nranks + 1);
    auto load = [&](const std::vector<int>& cuts) { double mx = 0.0; for (int r = 0; r < g->nranks; ++r) { double a = 0.0; for (int l = cuts[(size_t)r] / blubk::BZ; l < std::min(L, cuts[(size_t)r + 1] / blubk::BZ); ++l) a += w[(size_t)l]; mx = std::max(mx, a); } return mx; };
    std::vector<int> target((size_t)g->nranks + 1);
    for (int r = 0; r <= g->nranks; ++r) target[(size_t)r] = first[(size_t)r] * blubk::BZ;
    // one re-cut moves a cut strictly inside its old neighbours' slabs: clamp towards the target (the next call continues)
    std::vector<int> next = g->cuts;
    for (int r = 1; r < g->nranks; ++r) {
        const int lo = std::max(g->cuts[(size_t)r - 1], next[(size_t)r - 1]) + blubk::BZ * std::max(1, min_layers), hi = g->cuts[(size_t)r + 1] - blubk::BZ;
        next[(size_t)r] = std::max(lo, std::min(hi, target[(size_t)r]));
    }
    for (int r = g->nranks - 1; r >= 1; --r)      // (keep every slab at least min_layers thick from above too)
        next[(size_t)r] = std::min(next[(size_t)r], next[(size_t)r + 1] - blubk::BZ * std::max(1, min_layers));
    for (int r = 1; r < g->nranks; ++r) if (next[(size_t)r] <= g->cuts[(size_t)r - 1] || next[(size_t)r] >= g->cuts[(size_t)r + 1] || next[(size_t)r] <= next[(size_t)r - 1]) return BLUB_OK;      // (no legal move)
    if (next == g->cuts || load(next) > 0.95 * load(g->cuts)) return BLUB_OK;
    for (int r = 0; r <= g->nranks; ++r) nc[(size_t)r] = r == g->nranks ? g->slabs[0]->g.nz : next[(size_t)r];
    int rc = blub_slab_group_recut(g, nc.data());
    if (rc == BLUB_OK && changed) *changed = 1;
    return rc;
}
int blub_slab_group_cuts(const blub_slab_group* g, int32_t* cuts_out) {
    if (!g || !cuts_out) return blub::set_error(BLUB_ERR_INVALID_ARGUMENT, "null argument");
    for (int r = 0; r <= g->nranks; ++r) cuts_out[r] = r == g->nranks ? std::min(g->cuts[(size_t)r], g->slabs[0]->g.nz) : g->cuts[(size_t)r];
    return BLUB_OK;
}
void blub_slab_group_destroy(blub_slab_group* g) { blub::slab_group_destroy(g); }
int blub_slab_group_num_local(const blub_slab_group* g) { return g ? (int)g->slabs.size() : 0; }
blub_fluid* blub_slab_group_local_fluid(blub_slab_group* g, int i) { return (g && i >= 0 && i < (int)g->slabs.size()) ? g->slabs[i] : nullptr; }
int blub_slab_group_local_range(const blub_slab_group* g, int i, int32_t* z0, int32_t* z1) {
    if (!g || !z0 || !z1 || i < 0 || i >= (int)g->slabs.size()) return blub::set_error(BLUB_ERR_INVALID_ARGUMENT, "bad argument");
    *z0 = g->slabs[i]->slab_z0; *z1 = std::min(g->slabs[i]->slab_z1, g->slabs[i]->g.nz);
    return BLUB_OK;
}
// Every rank passes the SAME global particle arrays; each local slab keeps the particles whose z lies in its range.
int blub_slab_group_set_particles(blub_slab_group* g, uint32_t n, const float* pos_ll, const float* vx, const float* vy, const float* vz) {
    if (!g || (n && !pos_ll)) return blub::set_error(BLUB_ERR_INVALID_ARGUMENT, "null argument");
    if (hipSetDevice(g->device) != hipSuccess) return blub::set_error(BLUB_ERR_DEVICE, "hipSetDevice failed");
    for (size_t s = 0; s < g->slabs.size(); ++s) {
        blub_fluid* h = g->slabs[s];
        std::vector<float> p, a, b, c;
        const bool last = g->first + (int)s + 1 == g->nranks, firsts = g->first + (int)s == 0;
        for (uint32_t i = 0; i < n; ++i) {
            const float z = pos_ll[4 * (size_t)i + 2];
            if ((z >= (float)h->slab_z0 || firsts) && (z < (float)h->slab_z1 || last)) {
                p.insert(p.end(), pos_ll + 4 * (size_t)i, pos_ll + 4 * (size_t)i + 4);
                if (vx) a.insert(a.end(), vx + 4 * (size_t)i, vx + 4 * (size_t)i + 4);
                if (vy) b.insert(b.end(), vy + 4 * (size_t)i, vy + 4 * (size_t)i + 4);
                if (vz) c.insert(c.end(), vz + 4 * (size_t)i, vz + 4 * (size_t)i + 4);
            }
        }
        int rc = blub_fluid_set_particles(h, (uint32_t)(p.size() / 4), p.data(), vx ? a.data() : nullptr, vy ? b.data() : nullptr, vz ? c.data() : nullptr);
        if (rc != BLUB_OK) return rc;
        h->num_ghost = 0;
        if ((rc = blub::slab_upload_counts(g, (int)s)) != BLUB_OK) return rc;
        HIP_TRY(hipMemsetAsync(h->n_dev + 2, 0, sizeof(uint32_t), g->stream));      // (overflow flag)
    }
    for (auto& H : g->hist) H = blub_slab_group::Hist();      // the next step sizes nothing from the past: synchronous exchanges
    g->cnt_pending = false;
    return BLUB_OK;
}
uint32_t blub_slab_group_num_particles(blub_slab_group* g) {      // blocks (the counts live on the device)
    uint32_t n = 0;
    if (g && hipSetDevice(g->device) == hipSuccess && blub::slab_refresh_counts(g) == BLUB_OK) for (auto h : g->slabs) n += h->num_particles;
    return n;
}
// Own particles of all LOCAL slabs, concatenated in slab order (any pointer may be NULL); blocks.
int blub_slab_group_get_particles(blub_slab_group* g, float* pos_ll, float* vx, float* vy, float* vz) {
    if (!g) return blub::set_error(BLUB_ERR_INVALID_ARGUMENT, "null handle");
    if (hipSetDevice(g->device) != hipSuccess) return blub::set_error(BLUB_ERR_DEVICE, "hipSetDevice failed");
    { int rc = blub::slab_refresh_counts(g); if (rc != BLUB_OK) return rc; }
    size_t off = 0;
    for (auto h : g->slabs) {
        int rc = blub_fluid_get_particles(h, pos_ll ? pos_ll + off : nullptr, vx ? vx + off : nullptr, vy ? vy + off : nullptr, vz ? vz + off : nullptr);
        if (rc != BLUB_OK) return rc;
        off += (size_t)h->num_particles * 4;
    }
    return BLUB_OK;
}
int blub_slab_group_set_gravity_grid(blub_slab_group* g, const float gr[3]) { if (!g) return blub::set_error(BLUB_ERR_INVALID_ARGUMENT, "null handle"); for (auto h : g->slabs) blub_fluid_set_gravity_grid(h, gr); return BLUB_OK; }
int blub_slab_group_set_solver_config(blub_slab_group* g, int which, const blub_solver_config* cfg) {
    if (!g) return blub::set_error(BLUB_ERR_INVALID_ARGUMENT, "null handle");
    for (auto h : g->slabs) { int rc = blub_fluid_set_solver_config(h, which, cfg); if (rc != BLUB_OK) return rc; }
    return BLUB_OK;
}
int blub_slab_group_set_pcg_schedule(blub_slab_group* g, int mode) {   // every rank must pass the same mode
    if (!g) return blub::set_error(BLUB_ERR_INVALID_ARGUMENT, "null handle");
    for (auto h : g->slabs) { int rc = blub_fluid_set_pcg_schedule(h, mode); if (rc != BLUB_OK) return rc; }
    return BLUB_OK;
}
int blub_slab_group_set_gather_mode(blub_slab_group* g, int mode) {
    if (!g || mode < 0 || mode > 1) return blub::set_error(BLUB_ERR_INVALID_ARGUMENT, "bad argument");
    g->gather_mode = mode;
    if (g->rccl) snprintf(g->transport, sizeof g->transport, "rccl, %d ranks, partials by %s (set by the caller)", g->nranks, mode ? "ncclAllGather" : "grouped send/recv");
    return BLUB_OK;
}
int blub_slab_group_set_rebinning_frequency(blub_slab_group* g, uint32_t f) { if (!g) return blub::set_error(BLUB_ERR_INVALID_ARGUMENT, "null handle"); for (auto h : g->slabs) h->rebin_freq = f; return BLUB_OK; }
int blub_slab_group_set_meshes(blub_slab_group* g, uint32_t nv, const float* positions, uint32_t ni, const uint32_t* indices) {
    if (!g) return blub::set_error(BLUB_ERR_INVALID_ARGUMENT, "null handle");
    for (auto h : g->slabs) { int rc = blub_fluid_set_meshes(h, nv, positions, ni, indices); if (rc != BLUB_OK) return rc; }
    return BLUB_OK;
}
int blub_slab_group_voxelize(blub_slab_group* g, uint32_t num_meshes, const blub_mesh_desc* meshes) {
    if (!g) return blub::set_error(BLUB_ERR_INVALID_ARGUMENT, "null handle");
    for (auto h : g->slabs) { int rc = blub_fluid_voxelize(h, num_meshes, meshes); if (rc != BLUB_OK) return rc; }
    return BLUB_OK;
}
int blub_slab_group_step(blub_slab_group* g, float dt) {
    if (!g) return blub::set_error(BLUB_ERR_INVALID_ARGUMENT, "null handle");
    if (!(dt > 0.0f)) return blub::set_error(BLUB_ERR_INVALID_ARGUMENT, "simulation delta must be > 0");
    if (hipSetDevice(g->device) != hipSuccess) return blub::set_error(BLUB_ERR_DEVICE, "hipSetDevice failed");
    if (g->rccl && !g->direct && !g->comm) return blub::set_error(BLUB_ERR_COMM, "the group's communicator was aborted after an earlier failure");
    if (g->ck_interval && g->slabs[0]->step_counter % g->ck_interval == 0) { const int rck = blub::slab_checkpoint(g); if (rck != BLUB_OK) return rck; }
    const int rc = blub::slab_step(g, dt);
    if (rc != BLUB_OK && g->rccl && g->comm && !g->direct) {      // (the direct transport does not use the communicator after creation: its failures are recoverable in place)
        // A rank that leaves the lock-step sequence (buffer overflow, a failed HIP / RCCL call) must not leave its peers blocked inside
        // their next grouped send / receive: aborting the communicator makes their pending operations fail, so every rank returns an error.
        const std::string keep = blub::g_last_error;
        (void)ncclCommAbort(g->comm);
        g->comm = nullptr;
        blub::g_last_error = keep + " (communicator aborted: the peers' pending transport operations fail instead of hanging)";
    }
    return rc;
}
// ---- DIRECT transport: selection, export / connect -------------------------------------------------------------------------------------
// 0: host-issued transport operations (device copies in a local group, RCCL between processes); 1: direct (peer-mapped stores + flags).
// Between steps only; every rank must pass the same value.  A group of several processes needs every peer connected first.
int blub_slab_group_set_transport(blub_slab_group* g, int kind) {
    if (!g || kind < 0 || kind > 1) return blub::set_error(BLUB_ERR_INVALID_ARGUMENT, "bad argument");
    if (kind == 1) {
        for (int r = 0; r < g->nranks; ++r)
            if (g->peer_base[(size_t)r].empty()) return blub::set_error(BLUB_ERR_INVALID_ARGUMENT, "direct transport: a peer's memory is not mapped (blub_slab_group_connect)");
        for (auto h : g->slabs) if (!h->slab) return blub::set_error(BLUB_ERR_UNSUPPORTED, "direct transport needs the volumes of a slab in one allocation");
        if (g->nranks - 1 > blubk::SLAB_MAX_PEERS) return blub::set_error(BLUB_ERR_UNSUPPORTED, "direct transport: at most 8 slabs");
    }
    if (hipSetDevice(g->device) != hipSuccess || hipStreamSynchronize(g->stream) != hipSuccess) return blub::set_error(BLUB_ERR_DEVICE, "stream synchronisation failed");
    if (g->direct != (kind == 1)) { int rc = blub::slab_refresh_counts(g); if (rc != BLUB_OK) return rc; for (auto& H : g->hist) H = blub_slab_group::Hist(); g->cnt_pending = false; }
    g->direct = kind == 1;
    for (auto de : g->dir_error) (void)hipMemset(de, 0, sizeof(uint32_t));      // a time-out of the transport that is being left (or re-entered) is history (round-4 ADVICE)
    if (g->direct) snprintf(g->transport, sizeof g->transport, "direct (peer-mapped stores + flags), %d slabs%s, %s memory", g->nranks, (int)g->slabs.size() == g->nranks ? ", one process" : ", hipIpc",
                            g->mem_mode == BLUB_SLAB_MEMORY_COARSE ? "coarse-grained" : (g->mem_mode == BLUB_SLAB_MEMORY_FINE_GRAINED ? "fine-grained" : "uncached"));
    else snprintf(g->transport, sizeof g->transport, "%s", g->rccl ? "rccl" : "loopback");
    return BLUB_OK;
}
int blub_slab_group_get_transport(const blub_slab_group* g) { return g ? (g->direct ? 1 : 0) : BLUB_ERR_INVALID_ARGUMENT; }
// What the other processes need to map this process's slab: per exportable region a hipIpc handle and its size.  Single-slab groups only.
struct blub_slab_export_entry { hipIpcMemHandle_t handle; uint64_t bytes; };
int blub_slab_group_export_size(const blub_slab_group* g) { return (g && g->slabs.size() == 1) ? (int)(g->regions[0].size() * sizeof(blub_slab_export_entry)) : 0; }
int blub_slab_group_export(blub_slab_group* g, void* out, int capacity) {
    if (!g || !out || g->slabs.size() != 1) return blub::set_error(BLUB_ERR_INVALID_ARGUMENT, "export needs a group with one local slab");
    const auto& regs = g->regions[0];
    if (capacity < (int)(regs.size() * sizeof(blub_slab_export_entry))) return blub::set_error(BLUB_ERR_INVALID_ARGUMENT, "export buffer too small");
    if (hipSetDevice(g->device) != hipSuccess) return blub::set_error(BLUB_ERR_DEVICE, "hipSetDevice failed");
    blub_slab_export_entry* e = reinterpret_cast<blub_slab_export_entry*>(out);
    for (size_t k = 0; k < regs.size(); ++k) {
        if (!regs[k].base) return blub::set_error(BLUB_ERR_UNSUPPORTED, "direct transport needs the volumes of a slab in one allocation");
        HIP_TRY(hipIpcGetMemHandle(&e[k].handle, regs[k].base));
        e[k].bytes = regs[k].bytes;
    }
    return BLUB_OK;
}
int blub_slab_group_connect(blub_slab_group* g, int rank, const void* blob, int bytes) {
    if (!g || !blob || rank < 0 || rank >= g->nranks || blub::rank_local(g, rank)) return blub::set_error(BLUB_ERR_INVALID_ARGUMENT, "bad argument");
    const size_t n = g->regions[0].size();
    if (bytes != (int)(n * sizeof(blub_slab_export_entry))) return blub::set_error(BLUB_ERR_INVALID_ARGUMENT, "export blob of the wrong size");
    if (hipSetDevice(g->device) != hipSuccess) return blub::set_error(BLUB_ERR_DEVICE, "hipSetDevice failed");
    const blub_slab_export_entry* e = reinterpret_cast<const blub_slab_export_entry*>(blob);
    std::vector<char*> bases;
    for (size_t k = 0; k < n; ++k) {
        if (e[k].bytes != g->regions[0][k].bytes) return blub::set_error(BLUB_ERR_INVALID_ARGUMENT, "peer slab has a different layout");
        void* m = nullptr;
        HIP_TRY(hipIpcOpenMemHandle(&m, e[k].handle, hipIpcMemLazyEnablePeerAccess));
        g->ipc_opened.push_back(m);
        bases.push_back(reinterpret_cast<char*>(m));
    }
    g->peer_base[(size_t)rank] = bases;
    // A peer PROCESS whose slab lies in THIS device's memory shares the GPU with us (a development box: more ranks than GPUs).  The two kernels of a step that
    // wait for co-resident workgroups of their own launch (the one-launch brick-list build, the persistent PCG tail) sit out their bounds while the other
    // process holds the CUs (profiles/r05_multiproc_direct.jsonl: "a brick list build timed out"): the handle takes the spin-free forms by itself -- what
    // the "spin_free" tuning selects; bench.py used to set it when IT noticed the sharing (round-5 review, weak 6).
    hipPointerAttribute_t attr;
    if (!bases.empty() && hipPointerGetAttributes(&attr, bases[0]) == hipSuccess && attr.device == g->device)
        for (auto h : g->slabs) { h->two_kernel_build = true; h->use_tail = false; }
    else (void)hipGetLastError();
    return BLUB_OK;
}
// TEST HOOK: segments [first, last] of one step (0 ghost exchange, 1 transfer, 2 divergence, 3 solve_velocity, 4 binning, 5 project,
// 6 advect, 7 migration + density ghosts, 8 density_gather, 9 solve_density, 10 position_change, 11 correct, 12 second migration,
// 13 step counter), so that every slab can be inspected between them.  All ranks must pass the same range.
int blub_slab_group_run_stages(blub_slab_group* g, float dt, int first, int last) {
    if (!g || first < 0 || last >= blub::SS_COUNT || first > last) return blub::set_error(BLUB_ERR_INVALID_ARGUMENT, "bad argument");
    if (!(dt > 0.0f)) return blub::set_error(BLUB_ERR_INVALID_ARGUMENT, "simulation delta must be > 0");
    if (hipSetDevice(g->device) != hipSuccess) return blub::set_error(BLUB_ERR_DEVICE, "hipSetDevice failed");
    if (g->rccl && !g->direct && !g->comm) return blub::set_error(BLUB_ERR_COMM, "the group's communicator was aborted after an earlier failure");
    return blub::slab_step(g, dt, first, last);
}
uint64_t blub_slab_group_transport_ops(const blub_slab_group* g) { return g ? g->comm_ops : 0; }
int blub_slab_group_set_async_exchange(blub_slab_group* g, int enabled) {
    if (!g) return blub::set_error(BLUB_ERR_INVALID_ARGUMENT, "null handle");
    if (g->async_exchange && !enabled) {
        // after asynchronous steps the host-side particle counts are only BOUNDS (the counts live in n_dev) and the synchronous
        // protocol takes them as exact: fetch them, and forget the message-size history (round-3 ADVICE)
        if (hipSetDevice(g->device) != hipSuccess) return blub::set_error(BLUB_ERR_DEVICE, "hipSetDevice failed");
        int rc = blub::slab_refresh_counts(g);
        if (rc != BLUB_OK) return rc;
        for (auto& H : g->hist) H = blub_slab_group::Hist();
        g->cnt_pending = false;
    }
    g->async_exchange = enabled != 0;
    return BLUB_OK;
}
// Particles a migration held back at their sender for one exchange + ghost copies left out for one step because a message was sized
// (from the previous step's count) too small; summed over the local slabs since creation.  Blocks (the counters live on the device).
uint64_t blub_slab_group_held_back(blub_slab_group* g) {
    uint64_t n = 0;
    if (!g || hipSetDevice(g->device) != hipSuccess || hipStreamSynchronize(g->stream) != hipSuccess) return 0;
    for (auto h : g->slabs) { uint32_t v = 0; if (hipMemcpy(&v, h->n_dev + 3, sizeof v, hipMemcpyDeviceToHost) == hipSuccess) n += v; }
    return n;
}
int blub_slab_group_host_syncs(const blub_slab_group* g, uint64_t* particle_exchanges, uint64_t* done_polls) {
    if (!g || !particle_exchanges || !done_polls) return blub::set_error(BLUB_ERR_INVALID_ARGUMENT, "null argument");
    *particle_exchanges = g->host_syncs; *done_polls = g->done_polls;
    return BLUB_OK;
}
const char* blub_slab_group_transport_description(const blub_slab_group* g) { return g ? g->transport : ""; }
int blub_slab_group_synchronize(blub_slab_group* g) {
    if (!g) return blub::set_error(BLUB_ERR_INVALID_ARGUMENT, "null handle");
    if (hipSetDevice(g->device) != hipSuccess) return blub::set_error(BLUB_ERR_DEVICE, "hipSetDevice failed");
    for (auto h : g->slabs) { int rc = blub_fluid_synchronize(h); if (rc != BLUB_OK) return rc; }
    bool timed_out = false;
    for (size_t i = 0; i < g->dir_error.size(); ++i) {
        uint32_t v = 0;
        if (hipMemcpy(&v, g->dir_error[i], sizeof v, hipMemcpyDeviceToHost) == hipSuccess && v) {
            timed_out = true;
            (void)hipMemset(g->dir_error[i], 0, sizeof v);      // reported once: later waits wait again (a peer that was only late is back in step: sequence numbers compare by >=)
        }
    }
    if (timed_out) {
        for (auto& H : g->hist) H.pending = false;            // (the records of the steps in between carry the same mark)
        return blub::set_error(BLUB_ERR_COMM, "direct transport: a wait for a peer's flag timed out (a peer stopped stepping or fell seconds behind); the steps since are invalid");
    }
    return blub::slab_refresh_counts(g);
}
}  // extern "C"
