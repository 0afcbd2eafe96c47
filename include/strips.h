// strips.h — C++ face of the row-strip / multi-GPU part of the C ABI (rmd_strip_* / rmd_halo_* / rmd_comm_* in
// rmd_api.h).  Nothing like it exists in the reference (one device, default stream: SURVEY §0.4).
//
//   StripPlan plan(height, world, rank, params);     which rows this rank owns / holds / exchanges
//   NodeDenoiser node(width, height, params, devices);   one process driving several GPUs of a node:
//       node.denoise(color, nd, motion, prevNd, out) runs every rank's strip on its own device and
//       stream and then the neighbour history-halo exchange (RCCL over xGMI; ranks that share a device
//       -- a rehearsal on one GPU -- copy the rows device-to-device by the same plan).  With
//       params.exchange_iteration = X >= 0 the frame additionally exchanges AX's halo rows in its middle
//       (AX's boundary rows first; they travel on a second stream per rank beside AX's interior rows) and runs T, V,
//       A0..AX on fewer rows.
// A multi-PROCESS host (one rank per process) uses StripPlan + SvgfContext + rmd_comm_create /
// rmd_halo_exchange directly: INTEGRATION.md shows the loop.
#ifndef RMD_STRIPS_H
#define RMD_STRIPS_H

#include <memory>
#include <vector>

#include "svgf.h"

struct StripPlan : rmd_strip_plan {
    StripPlan() : rmd_strip_plan{} {}
    StripPlan(int height, int world, int rank, const SvgfParams& p) { rmdCheck(rmd_strip_plan_make(height, world, rank, &p, this), "StripPlan"); }
    std::vector<rmd_halo_step> haloSteps() const
    {
        std::vector<rmd_halo_step> s(RMD_HALO_MAX_STEPS);
        int n = 0;
        rmdCheck(rmd_halo_plan(this, s.data(), (int)s.size(), &n), "StripPlan::haloSteps");
        s.resize(n);
        return s;
    }
    size_t haloBytes(int width) const { return rmd_halo_bytes(this, width); }
    // the exchange INSIDE a frame (params.exchange_iteration >= 0): mid_rows rows of iteration mid_iteration's output per side
    std::vector<rmd_halo_step> midSteps() const
    {
        std::vector<rmd_halo_step> s(RMD_HALO_MAX_STEPS);
        int n = 0;
        rmdCheck(rmd_mid_halo_plan(this, s.data(), (int)s.size(), &n), "StripPlan::midSteps");
        s.resize(n);
        return s;
    }
};

// RAII over rmd_comm (move-only)
class Communicator {
    rmd_comm* c = nullptr;
public:
    Communicator() = default;
    explicit Communicator(const std::vector<int>& devices) { rmdCheck(rmd_comm_create_all((int)devices.size(), devices.data(), &c), "Communicator"); }
    Communicator(const void* id128, int world, int rank) { rmdCheck(rmd_comm_create(id128, world, rank, &c), "Communicator"); }
    Communicator(const Communicator&) = delete;
    Communicator& operator=(const Communicator&) = delete;
    Communicator(Communicator&& o) noexcept : c(o.c) { o.c = nullptr; }
    Communicator& operator=(Communicator&& o) noexcept { if (this != &o) { rmd_comm_destroy(c); c = o.c; o.c = nullptr; } return *this; }
    ~Communicator() { rmd_comm_destroy(c); }
    rmd_comm* get() const { return c; }
};

// One process, `devices.size()` ranks: rank k runs on devices[k].  The planes handed to denoise() are per
// rank and hold that rank's buffer rows [plan.buf_row0, plan.buf_row0 + plan.buf_rows).
class NodeDenoiser {
public:
    struct Rank {
        int device = 0;
        StripPlan plan;
        std::unique_ptr<SvgfContext> ctx;
        void* stream = nullptr;
        void* commStream = nullptr;                  // carries the mid-frame exchange beside the interior rows of the exchanged iteration
        void* headDone = nullptr; void* midDone = nullptr;   // events
    };
    std::vector<Rank> ranks;

    NodeDenoiser(int width, int height, const SvgfParams& p, const std::vector<int>& devices) : width_(width), params_(p)
    {
        const int world = (int)devices.size();
        bool distinct = true;
        for (int a = 0; a < world; ++a) for (int b = a + 1; b < world; ++b) distinct = distinct && devices[a] != devices[b];
        ranks.resize(world);
        for (int k = 0; k < world; ++k) {
            Rank& r = ranks[k];
            r.device = devices[k];
            r.plan = StripPlan(height, world, k, p);
            rmdCheck(rmd_set_device(r.device), "NodeDenoiser(set device)");
            r.ctx.reset(new SvgfContext(width, height, r.plan.buf_row0, r.plan.buf_rows));
            rmdCheck(rmd_stream_create(&r.stream), "NodeDenoiser(stream)");
            rmdCheck(rmd_stream_create(&r.commStream), "NodeDenoiser(comm stream)");
            rmdCheck(rmd_event_create(&r.headDone), "NodeDenoiser(event)");
            rmdCheck(rmd_event_create(&r.midDone), "NodeDenoiser(event)");
        }
        if (world > 1 && distinct) comm_ = Communicator(devices);      // RCCL: one communicator per device
    }
    NodeDenoiser(const NodeDenoiser&) = delete;
    NodeDenoiser& operator=(const NodeDenoiser&) = delete;
    ~NodeDenoiser()
    {
        for (Rank& r : ranks) {
            rmd_set_device(r.device);
            rmd_stream_sync(r.stream); rmd_stream_sync(r.commStream);
            rmd_event_destroy(r.headDone); rmd_event_destroy(r.midDone);
            rmd_stream_destroy(r.commStream); rmd_stream_destroy(r.stream);
            r.ctx.reset();
        }
    }

    // One frame: every rank's strip (asynchronous on its own stream), then the history halo for the next frame.
    void denoise(const std::vector<const float*>& color, const std::vector<const float*>& nd, const std::vector<const float*>& motion,
                 const std::vector<const float*>& prevNd, const std::vector<float*>& out)
    {
        const int world = (int)ranks.size();
        const bool mid = world > 1 && ranks[0].plan.mid_iteration >= 0;
        if (!mid) {
            for (int k = 0; k < world; ++k) {
                Rank& r = ranks[k];
                rmdCheck(rmd_set_device(r.device), "NodeDenoiser::denoise(set device)");
                r.ctx->denoise(params_, color[k], nd[k], motion[k], prevNd[k], out[k], r.plan.row0, r.plan.row1, r.stream);
            }
        } else {
            // One neighbour exchange inside the frame: every rank's T + V + A0..A(X-1) + the boundary rows of AX, which then
            // travel on the comm streams while AX runs on its interior rows; A(X+1) ... wait for the halo.
            auto part = [&](int k, int which) {
                Rank& r = ranks[k];
                rmdCheck(rmd_set_device(r.device), "NodeDenoiser::denoise(set device)");
                r.ctx->denoisePart(params_, color[k], nd[k], motion[k], prevNd[k], out[k], r.plan.row0, r.plan.row1, r.stream, which);
            };
            std::vector<float*> planes(world);
            for (int k = 0; k < world; ++k) {
                part(k, RMD_ATROUS_HEAD);
                planes[k] = ranks[k].ctx->midPlane(params_);
                rmdCheck(rmd_event_record(ranks[k].headDone, ranks[k].stream), "NodeDenoiser(head event)");
                rmdCheck(rmd_stream_wait_event(ranks[k].commStream, ranks[k].headDone), "NodeDenoiser(comm waits for head)");
            }
            if (comm_.get()) {
                std::vector<rmd_strip_plan> plans(world);
                std::vector<void*> streams(world);
                for (int k = 0; k < world; ++k) { plans[k] = ranks[k].plan; streams[k] = ranks[k].commStream; }
                rmdCheck(rmd_mid_exchange_all(comm_.get(), plans.data(), width_, planes.data(), streams.data()), "NodeDenoiser(mid exchange)");
            } else {
                // ranks share a device (rehearsal): a receiver's comm stream also waits for the SENDER's head, then copies by plan
                for (int k = 0; k < world; ++k)
                    for (const rmd_halo_step& s : ranks[k].plan.midSteps()) {
                        if (s.kind != RMD_HALO_RECV) continue;
                        const Rank& q = ranks[s.peer];
                        rmdCheck(rmd_stream_wait_event(ranks[k].commStream, q.headDone), "NodeDenoiser(comm waits for the sender)");
                        float* dst = planes[k] + (size_t)(s.row_lo - ranks[k].plan.buf_row0) * width_ * 4;
                        const float* src = planes[s.peer] + (size_t)(s.row_lo - q.plan.buf_row0) * width_ * 4;
                        rmdCheck(rmd_memcpy_d2d(dst, src, (size_t)(s.row_hi - s.row_lo) * width_ * 16, ranks[k].commStream), "NodeDenoiser(mid copy)");
                    }
            }
            for (int k = 0; k < world; ++k) {
                rmdCheck(rmd_set_device(ranks[k].device), "NodeDenoiser::denoise(set device)");
                rmdCheck(rmd_event_record(ranks[k].midDone, ranks[k].commStream), "NodeDenoiser(mid event)");
            }
            for (int k = 0; k < world; ++k) part(k, RMD_ATROUS_INTERIOR);
            for (int k = 0; k < world; ++k) {
                rmdCheck(rmd_set_device(ranks[k].device), "NodeDenoiser::denoise(set device)");
                rmdCheck(rmd_stream_wait_event(ranks[k].stream, ranks[k].midDone), "NodeDenoiser(wait for the halo)");
                // shared-device rehearsal: the neighbours' copies READ this rank's rows on THEIR comm streams; the next frame
                // must not overwrite them earlier (with RCCL the sends are on this rank's own comm stream: midDone covers them)
                if (!comm_.get())
                    for (int q : { k - 1, k + 1 })
                        if (q >= 0 && q < world) rmdCheck(rmd_stream_wait_event(ranks[k].stream, ranks[q].midDone), "NodeDenoiser(wait for the neighbour's copy)");
                part(k, RMD_ATROUS_TAIL);
            }
        }
        if (world == 1) return;
        std::vector<float*> hc(world), hm(world);
        std::vector<unsigned char*> hl(world);
        for (int k = 0; k < world; ++k) rmdCheck(rmd_svgf_context_history(ranks[k].ctx->get(), &hc[k], &hm[k], &hl[k]), "NodeDenoiser(history)");
        if (comm_.get()) {
            std::vector<rmd_strip_plan> plans(world);
            std::vector<void*> streams(world);
            for (int k = 0; k < world; ++k) { plans[k] = ranks[k].plan; streams[k] = ranks[k].stream; }
            rmdCheck(rmd_halo_exchange_all(comm_.get(), plans.data(), width_, hc.data(), hm.data(), hl.data(), streams.data()), "NodeDenoiser(exchange)");
            return;
        }
        // ranks share a device (rehearsal): the same plan, rows copied device to device once every strip is done
        for (Rank& r : ranks) rmdCheck(rmd_stream_sync(r.stream), "NodeDenoiser(sync)");
        for (int k = 0; k < world; ++k)
            for (const rmd_halo_step& s : ranks[k].plan.haloSteps()) {
                if (s.kind != RMD_HALO_RECV) continue;
                const Rank& q = ranks[s.peer];
                // hist_color float4, hist_moments float2, hist_len uint8 (include/rmd_api.h RMD_PLANE_*)
                const size_t px = s.plane == RMD_PLANE_HIST_COLOR ? 16 : s.plane == RMD_PLANE_HIST_MOMENTS ? 8 : 1;
                auto base = [&](int rank) { return s.plane == RMD_PLANE_HIST_COLOR ? (unsigned char*)hc[rank] : s.plane == RMD_PLANE_HIST_MOMENTS ? (unsigned char*)hm[rank] : hl[rank]; };
                unsigned char* dst = base(k) + (size_t)(s.row_lo - ranks[k].plan.buf_row0) * width_ * px;
                const unsigned char* src = base(s.peer) + (size_t)(s.row_lo - q.plan.buf_row0) * width_ * px;
                rmdCheck(rmd_memcpy_d2d(dst, src, (size_t)(s.row_hi - s.row_lo) * width_ * px, ranks[k].stream), "NodeDenoiser(copy)");
            }
    }
    void synchronize()
    {
        for (Rank& r : ranks) { rmdCheck(rmd_set_device(r.device), "NodeDenoiser::synchronize"); rmdCheck(rmd_stream_sync(r.stream), "NodeDenoiser::synchronize"); }
    }

private:
    int width_;
    SvgfParams params_;
    Communicator comm_;
};

#endif
