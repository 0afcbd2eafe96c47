// strips.hip — row-strip sharding of one frame across the GPUs of a node behind the C ABI
// (nothing like it in the reference, which is single-device: SURVEY §0.4).
//
//   rmd_strip_plan_make / rmd_halo_plan / rmd_halo_bytes   pure arithmetic, no device: which rows a
//       rank owns and holds, and which history rows travel to / from rank +-1 every frame.  The same
//       rules as raymarchdenoisercuda_amd/sharding.py (tests/test_strips_abi.py compares the two).
//   rmd_comm_*            an RCCL communicator: one per process and GPU (rmd_comm_create, unique id
//       handed over by the host program) or all GPUs of one process (rmd_comm_create_all).
//   rmd_halo_exchange     the per-frame exchange: ncclGroupStart; ncclSend / ncclRecv with rank +-1
//       (one direct xGMI link per neighbour); ncclGroupEnd -- on the caller's stream.
//
// librccl.so is opened on first use (dlopen), so librmd.so loads on machines without RCCL and a
// process that already carries an RCCL (PyTorch bundles one) keeps using that copy.
#include "common.h"
#include <dlfcn.h>
#include <cstring>
#include <mutex>
#include <vector>

namespace rmd {

// ---- the handful of RCCL entry points used, with their documented C signatures (rccl.h) ----------
typedef struct ncclComm* ncclComm_t;
typedef struct { char internal[128]; } ncclUniqueId;
enum { kNcclSuccess = 0, kNcclUint8 = 1, kNcclFloat32 = 7 };     // rccl.h ncclResult_t / ncclDataType_t
struct Rccl {
    void* handle = nullptr;
    int (*GetUniqueId)(ncclUniqueId*) = nullptr;
    int (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    int (*CommInitAll)(ncclComm_t*, int, const int*) = nullptr;
    int (*CommDestroy)(ncclComm_t) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    int (*Send)(const void*, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
    int (*Recv)(void*, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
};

static Rccl* rccl()
{
    static Rccl lib;
    static std::once_flag once;
    std::call_once(once, [] {
        const char* names[] = { getenv("RMD_RCCL_PATH"), "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1" };
        for (const char* n : names) {
            if (!n) continue;
            lib.handle = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
            if (lib.handle) break;
        }
        if (!lib.handle) return;
        auto sym = [&](const char* s) { return dlsym(lib.handle, s); };
        lib.GetUniqueId = reinterpret_cast<decltype(lib.GetUniqueId)>(sym("ncclGetUniqueId"));
        lib.CommInitRank = reinterpret_cast<decltype(lib.CommInitRank)>(sym("ncclCommInitRank"));
        lib.CommInitAll = reinterpret_cast<decltype(lib.CommInitAll)>(sym("ncclCommInitAll"));
        lib.CommDestroy = reinterpret_cast<decltype(lib.CommDestroy)>(sym("ncclCommDestroy"));
        lib.GroupStart = reinterpret_cast<decltype(lib.GroupStart)>(sym("ncclGroupStart"));
        lib.GroupEnd = reinterpret_cast<decltype(lib.GroupEnd)>(sym("ncclGroupEnd"));
        lib.Send = reinterpret_cast<decltype(lib.Send)>(sym("ncclSend"));
        lib.Recv = reinterpret_cast<decltype(lib.Recv)>(sym("ncclRecv"));
        lib.GetErrorString = reinterpret_cast<decltype(lib.GetErrorString)>(sym("ncclGetErrorString"));
        if (!lib.GetUniqueId || !lib.CommInitRank || !lib.CommInitAll || !lib.CommDestroy || !lib.GroupStart || !lib.GroupEnd ||
            !lib.Send || !lib.Recv) {
            dlclose(lib.handle);
            lib.handle = nullptr;
        }
    });
    return lib.handle ? &lib : nullptr;
}

static int nccl_fail(Rccl* r, int code, const char* what)
{
    return fail(RMD_E_COMM, "%s: %s", what, (r && r->GetErrorString) ? r->GetErrorString(code) : "RCCL error");
}

#define RMD_NCCL(r, call)                                             \
    do {                                                              \
        const int rmd_n_ = (call);                                    \
        if (rmd_n_ != kNcclSuccess) return nccl_fail(r, rmd_n_, #call); \
    } while (0)

}  // namespace rmd

using namespace rmd;

struct rmd_comm {
    int world = 0;
    std::vector<ncclComm_t> comms;     // one (multi-process) or `world` of them (single process)
    std::vector<int> devices;          // device of comms[i] (single-process form), else empty
};

extern "C" {

// ---- planning ------------------------------------------------------------------------------------
int rmd_strip_rows(int height, int world, int rank, int* row0, int* row1)
{
    if (!row0 || !row1) return fail(RMD_E_NULL, "rmd_strip_rows: row0/row1 is NULL");
    if (height <= 0 || world <= 0 || rank < 0 || rank >= world)
        return fail(RMD_E_PARAM, "rmd_strip_rows: height %d, world %d, rank %d", height, world, rank);
    const int base = height / world, rem = height % world;
    *row0 = rank * base + (rank < rem ? rank : rem);
    *row1 = *row0 + base + (rank < rem ? 1 : 0);
    return RMD_OK;
}

int rmd_strip_plan_make(int height, int world, int rank, const rmd_svgf_params* p, rmd_strip_plan* out)
{
    if (!p || !out) return fail(RMD_E_NULL, "rmd_strip_plan_make: params/out is NULL");
    int reach[4], mid[2];
    if (int e = rmd_svgf_frame_reach(p, reach)) return e;
    if (int e = rmd_svgf_frame_mid_exchange(p, mid)) return e;
    int row0, row1;
    if (int e = rmd_strip_rows(height, world, rank, &row0, &row1)) return e;
    if (world > 1 && height / world < reach[1])
        return fail(RMD_E_ROWS, "rmd_strip_plan_make: strips of %d rows are shorter than the history reach %d: use fewer ranks, "
                    "a taller frame or a smaller max_motion_rows", height / world, reach[1]);
    if (world > 1 && height / world < mid[1])
        return fail(RMD_E_ROWS, "rmd_strip_plan_make: strips of %d rows are shorter than the %d rows of the mid-frame exchange: use fewer "
                    "ranks or a later exchange_iteration", height / world, mid[1]);
    int r = reach[0] > reach[1] ? reach[0] : reach[1];
    if (mid[1] > r) r = mid[1];
    const int b0 = row0 - r > 0 ? row0 - r : 0, b1 = row1 + r < height ? row1 + r : height;
    out->height = height; out->world = world; out->rank = rank;
    out->row0 = row0; out->row1 = row1;
    out->buf_row0 = b0; out->buf_rows = b1 - b0;
    out->reach_in = reach[0]; out->reach_hist = reach[1]; out->have_color = reach[2]; out->have_moments = reach[3];
    out->mid_iteration = mid[0]; out->mid_rows = mid[1];
    return RMD_OK;
}

// The exchange INSIDE a frame (rmd_svgf_params.exchange_iteration): the mid_rows rows of iteration X's output beyond either
// end of the strip come from the neighbour that computed them as its own rows.  plane index RMD_PLANE_MID.
int rmd_mid_halo_plan(const rmd_strip_plan* plan, rmd_halo_step* steps, int max_steps, int* n_steps)
{
    if (!plan || !n_steps) return fail(RMD_E_NULL, "rmd_mid_halo_plan: plan/n_steps is NULL");
    int n = 0;
    auto push = [&](int kind, int lo, int hi, int peer) {
        if (hi <= lo) return;
        if (steps && n < max_steps) steps[n] = rmd_halo_step{ kind, RMD_PLANE_MID, lo, hi, peer };
        ++n;
    };
    auto clamp = [&](int v) { return v < 0 ? 0 : (v > plan->height ? plan->height : v); };
    if (plan->world > 1 && plan->mid_iteration >= 0 && plan->mid_rows > 0) {
        const int up = plan->rank - 1, down = plan->rank + 1, R = plan->mid_rows;
        if (up >= 0) {
            push(RMD_HALO_RECV, clamp(plan->row0 - R), plan->row0, up);
            push(RMD_HALO_SEND, plan->row0, clamp(plan->row0 + R), up);
        }
        if (down < plan->world) {
            push(RMD_HALO_RECV, plan->row1, clamp(plan->row1 + R), down);
            push(RMD_HALO_SEND, clamp(plan->row1 - R), plan->row1, down);
        }
    }
    *n_steps = n;
    if (steps && n > max_steps) return fail(RMD_E_BUFFER, "rmd_mid_halo_plan: %d steps, room for %d", n, max_steps);
    return RMD_OK;
}

int rmd_halo_plan(const rmd_strip_plan* plan, rmd_halo_step* steps, int max_steps, int* n_steps)
{
    if (!plan || !n_steps) return fail(RMD_E_NULL, "rmd_halo_plan: plan/n_steps is NULL");
    int n = 0;
    auto push = [&](int kind, int plane, int lo, int hi, int peer) {
        if (hi <= lo) return;
        if (steps && n < max_steps) steps[n] = rmd_halo_step{ kind, plane, lo, hi, peer };
        ++n;
    };
    auto clamp = [&](int v) { return v < 0 ? 0 : (v > plan->height ? plan->height : v); };
    if (plan->world > 1) {
        const int up = plan->rank - 1, down = plan->rank + 1, R = plan->reach_hist;
        // hist_color; hist_moments; hist_len (T writes the moments and the history length on the same rows)
        const int planes[3] = { RMD_PLANE_HIST_COLOR, RMD_PLANE_HIST_MOMENTS, RMD_PLANE_HIST_LEN };
        const int have[3] = { plan->have_color, plan->have_moments, plan->have_moments };
        for (int q = 0; q < 3; ++q) {
            const int pl = planes[q];
            if (have[q] >= R) continue;
            if (up >= 0) {
                push(RMD_HALO_RECV, pl, clamp(plan->row0 - R), clamp(plan->row0 - have[q]), up);
                // rank-1's lower need [row0+have, row0+R) lies in this rank's strip
                push(RMD_HALO_SEND, pl, clamp(plan->row0 + have[q]), clamp(plan->row0 + R), up);
            }
            if (down < plan->world) {
                push(RMD_HALO_RECV, pl, clamp(plan->row1 + have[q]), clamp(plan->row1 + R), down);
                push(RMD_HALO_SEND, pl, clamp(plan->row1 - R), clamp(plan->row1 - have[q]), down);
            }
        }
    }
    *n_steps = n;
    if (steps && n > max_steps) return fail(RMD_E_BUFFER, "rmd_halo_plan: %d steps, room for %d", n, max_steps);
    return RMD_OK;
}

// bytes per pixel of the planes a halo step may name (include/rmd_api.h RMD_PLANE_*)
static size_t plane_pixel_bytes(int plane)
{
    return plane == RMD_PLANE_HIST_MOMENTS ? 8u : plane == RMD_PLANE_HIST_LEN ? 1u : 16u;
}

size_t rmd_halo_bytes(const rmd_strip_plan* plan, int width)
{
    if (!plan || width <= 0) return 0;
    rmd_halo_step st[RMD_HALO_MAX_STEPS];
    int n = 0;
    if (rmd_halo_plan(plan, st, RMD_HALO_MAX_STEPS, &n) != RMD_OK) return 0;
    size_t total = 0;
    for (int i = 0; i < n; ++i)
        if (st[i].kind == RMD_HALO_RECV) total += (size_t)(st[i].row_hi - st[i].row_lo) * (size_t)width * plane_pixel_bytes(st[i].plane);
    return total;
}

// ---- RCCL ----------------------------------------------------------------------------------------
int rmd_comm_available(void) { return rccl() ? 1 : 0; }

int rmd_comm_unique_id(void* id128)
{
    if (!id128) return fail(RMD_E_NULL, "rmd_comm_unique_id: id is NULL");
    Rccl* r = rccl();
    if (!r) return fail(RMD_E_COMM, "rmd_comm_unique_id: librccl.so not found (set RMD_RCCL_PATH)");
    ncclUniqueId id;
    RMD_NCCL(r, r->GetUniqueId(&id));
    memcpy(id128, &id, sizeof(id));
    return RMD_OK;
}

int rmd_comm_create(const void* id128, int world, int rank, rmd_comm** out)
{
    if (!id128 || !out) return fail(RMD_E_NULL, "rmd_comm_create: id/out is NULL");
    if (world <= 0 || rank < 0 || rank >= world) return fail(RMD_E_PARAM, "rmd_comm_create: world %d, rank %d", world, rank);
    Rccl* r = rccl();
    if (!r) return fail(RMD_E_COMM, "rmd_comm_create: librccl.so not found (set RMD_RCCL_PATH)");
    ncclUniqueId id;
    memcpy(&id, id128, sizeof(id));
    ncclComm_t c = nullptr;
    RMD_NCCL(r, r->CommInitRank(&c, world, id, rank));     // on the calling thread's current device
    rmd_comm* m = new rmd_comm;
    m->world = world;
    m->comms.push_back(c);
    *out = m;
    return RMD_OK;
}

int rmd_comm_create_all(int ndev, const int* devices, rmd_comm** out)
{
    if (!out) return fail(RMD_E_NULL, "rmd_comm_create_all: out is NULL");
    if (ndev <= 0) return fail(RMD_E_PARAM, "rmd_comm_create_all: ndev %d", ndev);
    Rccl* r = rccl();
    if (!r) return fail(RMD_E_COMM, "rmd_comm_create_all: librccl.so not found (set RMD_RCCL_PATH)");
    rmd_comm* m = new rmd_comm;
    m->world = ndev;
    m->comms.resize(ndev);
    m->devices.resize(ndev);
    for (int i = 0; i < ndev; ++i) m->devices[i] = devices ? devices[i] : i;
    const int e = r->CommInitAll(m->comms.data(), ndev, m->devices.data());
    if (e != kNcclSuccess) { delete m; return nccl_fail(r, e, "ncclCommInitAll"); }
    *out = m;
    return RMD_OK;
}

int rmd_comm_destroy(rmd_comm* c)
{
    if (!c) return RMD_OK;
    Rccl* r = rccl();
    if (r) for (ncclComm_t k : c->comms) if (k) r->CommDestroy(k);
    delete c;
    return RMD_OK;
}

// One rank's part of an exchange: its steps, on its planes, inside an open group.
static int post_steps(Rccl* r, ncclComm_t comm, const rmd_halo_step* steps, int n, int buf_row0, int buf_rows, int width,
                      void* const planes[RMD_PLANE_COUNT], hipStream_t stream)
{
    for (int i = 0; i < n; ++i) {
        const rmd_halo_step& s = steps[i];
        if (s.plane < 0 || s.plane >= RMD_PLANE_COUNT) return fail(RMD_E_PARAM, "rmd_halo_exchange: plane index %d", s.plane);
        unsigned char* plane = static_cast<unsigned char*>(planes[s.plane]);
        if (!plane) return fail(RMD_E_NULL, "rmd_halo_exchange: plane %d is NULL", s.plane);
        if (s.row_lo < buf_row0 || s.row_hi > buf_row0 + buf_rows || s.row_lo >= s.row_hi)
            return fail(RMD_E_ROWS, "rmd_halo_exchange: rows [%d,%d) outside the planes [%d,%d)", s.row_lo, s.row_hi, buf_row0, buf_row0 + buf_rows);
        const size_t px = plane_pixel_bytes(s.plane);
        unsigned char* p = plane + (size_t)(s.row_lo - buf_row0) * (size_t)width * px;
        const size_t bytes = (size_t)(s.row_hi - s.row_lo) * (size_t)width * px;
        // float planes travel as floats, the history length plane as bytes
        const int type = px == 1 ? kNcclUint8 : kNcclFloat32;
        const size_t count = px == 1 ? bytes : bytes / 4;
        if (s.kind == RMD_HALO_SEND) RMD_NCCL(r, r->Send(p, count, type, s.peer, comm, stream));
        else                         RMD_NCCL(r, r->Recv(p, count, type, s.peer, comm, stream));
    }
    return RMD_OK;
}

int rmd_exchange_steps(rmd_comm* c, int comm_index, const rmd_halo_step* steps, int n_steps, int buf_row0, int buf_rows,
                       int width, void* const planes[RMD_PLANE_COUNT], void* stream)
{
    if (!c) return fail(RMD_E_NULL, "rmd_halo_exchange: communicator is NULL");
    if (comm_index < 0 || comm_index >= (int)c->comms.size()) return fail(RMD_E_PARAM, "rmd_halo_exchange: communicator index %d", comm_index);
    if (n_steps == 0) return RMD_OK;
    if (!steps || !planes || width <= 0) return fail(RMD_E_NULL, "rmd_halo_exchange: steps / planes is NULL or width <= 0");
    Rccl* r = rccl();
    if (!r) return fail(RMD_E_COMM, "rmd_halo_exchange: librccl.so not found");
    RMD_NCCL(r, r->GroupStart());
    const int e = post_steps(r, c->comms[comm_index], steps, n_steps, buf_row0, buf_rows, width, planes, as_stream(stream));
    const int g = r->GroupEnd();
    if (e) return e;
    if (g != kNcclSuccess) return nccl_fail(r, g, "ncclGroupEnd");
    return RMD_OK;
}

int rmd_halo_exchange_steps(rmd_comm* c, int comm_index, const rmd_halo_step* steps, int n_steps, int buf_row0, int buf_rows,
                            int width, float* hist_color, float* hist_moments, unsigned char* hist_len, void* stream)
{
    void* const planes[RMD_PLANE_COUNT] = { hist_color, hist_moments, nullptr, hist_len };
    return rmd_exchange_steps(c, comm_index, steps, n_steps, buf_row0, buf_rows, width, planes, stream);
}

int rmd_halo_exchange(rmd_comm* c, const rmd_strip_plan* plan, int width, float* hist_color, float* hist_moments,
                      unsigned char* hist_len, void* stream)
{
    if (!plan) return fail(RMD_E_NULL, "rmd_halo_exchange: plan is NULL");
    if (c && (int)c->comms.size() != 1) return fail(RMD_E_PARAM, "rmd_halo_exchange: one communicator per process expected; use rmd_halo_exchange_all");
    rmd_halo_step st[RMD_HALO_MAX_STEPS];
    int n = 0;
    if (int e = rmd_halo_plan(plan, st, RMD_HALO_MAX_STEPS, &n)) return e;
    if (n == 0) return RMD_OK;
    if (c && c->world != plan->world) return fail(RMD_E_PARAM, "rmd_halo_exchange: communicator of %d ranks, plan of %d", c->world, plan->world);
    return rmd_halo_exchange_steps(c, 0, st, n, plan->buf_row0, plan->buf_rows, width, hist_color, hist_moments, hist_len, stream);
}

int rmd_mid_exchange(rmd_comm* c, const rmd_strip_plan* plan, int width, float* mid_plane, void* stream)
{
    if (!plan) return fail(RMD_E_NULL, "rmd_mid_exchange: plan is NULL");
    if (c && (int)c->comms.size() != 1) return fail(RMD_E_PARAM, "rmd_mid_exchange: one communicator per process expected; use rmd_mid_exchange_all");
    rmd_halo_step st[RMD_HALO_MAX_STEPS];
    int n = 0;
    if (int e = rmd_mid_halo_plan(plan, st, RMD_HALO_MAX_STEPS, &n)) return e;
    if (n == 0) return RMD_OK;
    if (!c) return fail(RMD_E_NULL, "rmd_mid_exchange: communicator is NULL");
    if (c->world != plan->world) return fail(RMD_E_PARAM, "rmd_mid_exchange: communicator of %d ranks, plan of %d", c->world, plan->world);
    if (!mid_plane || width <= 0) return fail(RMD_E_NULL, "rmd_mid_exchange: plane is NULL or width <= 0");
    Rccl* r = rccl();
    if (!r) return fail(RMD_E_COMM, "rmd_mid_exchange: librccl.so not found");
    RMD_NCCL(r, r->GroupStart());
    void* const planes[RMD_PLANE_COUNT] = { nullptr, nullptr, mid_plane, nullptr };
    const int e = post_steps(r, c->comms[0], st, n, plan->buf_row0, plan->buf_rows, width, planes, as_stream(stream));
    const int g = r->GroupEnd();
    if (e) return e;
    if (g != kNcclSuccess) return nccl_fail(r, g, "ncclGroupEnd");
    return RMD_OK;
}

int rmd_mid_exchange_all(rmd_comm* c, const rmd_strip_plan* plans, int width, float* const* mid_planes, void* const* streams)
{
    if (!c || !plans || !mid_planes) return fail(RMD_E_NULL, "rmd_mid_exchange_all: NULL argument");
    if ((int)c->comms.size() != c->world) return fail(RMD_E_PARAM, "rmd_mid_exchange_all: needs a communicator from rmd_comm_create_all");
    Rccl* r = rccl();
    if (!r) return fail(RMD_E_COMM, "rmd_mid_exchange_all: librccl.so not found");
    int prev = 0;
    RMD_HIP(hipGetDevice(&prev));
    RMD_NCCL(r, r->GroupStart());
    int err = RMD_OK;
    for (int k = 0; k < c->world && !err; ++k) {
        rmd_halo_step st[RMD_HALO_MAX_STEPS];
        int n = 0;
        err = rmd_mid_halo_plan(&plans[k], st, RMD_HALO_MAX_STEPS, &n);
        if (err) break;
        if (hipSetDevice(c->devices[k]) != hipSuccess) { err = fail(RMD_E_PARAM, "rmd_mid_exchange_all: hipSetDevice(%d) failed", c->devices[k]); break; }
        void* const planes[RMD_PLANE_COUNT] = { nullptr, nullptr, mid_planes[k], nullptr };
        err = post_steps(r, c->comms[k], st, n, plans[k].buf_row0, plans[k].buf_rows, width, planes, as_stream(streams ? streams[k] : nullptr));
    }
    const int g = r->GroupEnd();
    (void)hipSetDevice(prev);
    if (err) return err;
    if (g != kNcclSuccess) return nccl_fail(r, g, "ncclGroupEnd");
    return RMD_OK;
}

int rmd_halo_exchange_all(rmd_comm* c, const rmd_strip_plan* plans, int width, float* const* hist_color, float* const* hist_moments,
                          unsigned char* const* hist_len, void* const* streams)
{
    if (!c || !plans || !hist_color || !hist_moments || !hist_len) return fail(RMD_E_NULL, "rmd_halo_exchange_all: NULL argument");
    if ((int)c->comms.size() != c->world) return fail(RMD_E_PARAM, "rmd_halo_exchange_all: needs a communicator from rmd_comm_create_all");
    Rccl* r = rccl();
    if (!r) return fail(RMD_E_COMM, "rmd_halo_exchange_all: librccl.so not found");
    int prev = 0;
    RMD_HIP(hipGetDevice(&prev));
    RMD_NCCL(r, r->GroupStart());
    int err = RMD_OK;
    for (int k = 0; k < c->world && !err; ++k) {
        rmd_halo_step st[RMD_HALO_MAX_STEPS];
        int n = 0;
        err = rmd_halo_plan(&plans[k], st, RMD_HALO_MAX_STEPS, &n);
        if (err) break;
        if (hipSetDevice(c->devices[k]) != hipSuccess) { err = fail(RMD_E_PARAM, "rmd_halo_exchange_all: hipSetDevice(%d) failed", c->devices[k]); break; }
        void* const planes[RMD_PLANE_COUNT] = { hist_color[k], hist_moments[k], nullptr, hist_len[k] };
        err = post_steps(r, c->comms[k], st, n, plans[k].buf_row0, plans[k].buf_rows, width, planes, as_stream(streams ? streams[k] : nullptr));
    }
    const int g = r->GroupEnd();
    (void)hipSetDevice(prev);
    if (err) return err;
    if (g != kNcclSuccess) return nccl_fail(r, g, "ncclGroupEnd");
    return RMD_OK;
}

}  // extern "C"
