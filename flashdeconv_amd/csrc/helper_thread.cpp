// One helper thread per process that queues launch sequences on behalf of an entry point (internal).
//
// A rank's share of a sharded fit is host-bound: the second phase of its shard build is ~35 dependent launches (0.12 ms of host
// time) that the sketch of its own rows does not depend on.  graph_shard_knn hands that phase to this thread and returns; the
// calling thread goes on to the sketch, and whoever needs the graph waits for the ticket (graph_meta_sync, fdx_shard_fit_dev).
// The thread only ever queues device work (pool allocations, launches, event records): it never waits for the device.
#include <condition_variable>
#include <deque>
#include <exception>
#include <mutex>
#include <thread>

#include "fdx_internal.h"

namespace fdx {

namespace {
struct Job { std::function<int()> fn; std::shared_ptr<HelperTicket> ticket; int dev; };
struct Helper {
    std::mutex mu;
    std::condition_variable cv;
    std::deque<Job> q;
    bool stop = false;
    std::thread th;
    void run() {
        for (;;) {
            Job job;
            {
                std::unique_lock<std::mutex> lk(mu);
                cv.wait(lk, [&] { return stop || !q.empty(); });
                if (q.empty()) return;
                job = std::move(q.front());
                q.pop_front();
            }
            int rc = hipSetDevice(job.dev) == hipSuccess ? 0 : FDX_ERR_HIP;
            std::string err = rc ? "helper thread: hipSetDevice failed" : "";
            if (!rc) {
                set_error("");
                try {
                    rc = job.fn();
                    if (rc) err = get_error();
                } catch (const std::exception& e) {       // e.g. bad_alloc: an error code for the waiting thread, not std::terminate
                    rc = FDX_ERR_INVALID;
                    err = std::string("helper thread: ") + e.what();
                }
            }
            {
                std::lock_guard<std::mutex> lk(job.ticket->mu);
                job.ticket->rc = rc;
                job.ticket->err = std::move(err);
                job.ticket->done = true;
            }
            job.ticket->cv.notify_all();
        }
    }
    ~Helper() {
        {
            std::lock_guard<std::mutex> lk(mu);
            stop = true;
        }
        cv.notify_all();
        if (th.joinable()) th.join();
    }
};
Helper& helper() {
    static Helper h;
    return h;
}
}  // namespace

std::shared_ptr<HelperTicket> helper_submit(std::function<int()> fn) {
    auto ticket = std::make_shared<HelperTicket>();
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) {
        (void)hipGetLastError();
        ticket->rc = fail(FDX_ERR_HIP, "helper thread: no current device");
        ticket->err = get_error();
        ticket->done = true;
        return ticket;
    }
    Helper& h = helper();
    bool queued = false;
    try {
        std::lock_guard<std::mutex> lk(h.mu);
        if (!h.th.joinable()) h.th = std::thread([&h] { h.run(); });   // may throw (thread limit of a container)
        h.q.push_back(Job{fn, ticket, dev});
        queued = true;
    } catch (...) {
    }
    if (!queued) {                                        // no helper: the caller does the work itself, now
        set_error("");
        ticket->rc = fn();
        if (ticket->rc) ticket->err = get_error();
        ticket->done = true;
        return ticket;
    }
    h.cv.notify_one();
    return ticket;
}

int helper_wait(const std::shared_ptr<HelperTicket>& ticket) {
    if (!ticket) return 0;
    std::unique_lock<std::mutex> lk(ticket->mu);
    ticket->cv.wait(lk, [&] { return ticket->done; });
    if (ticket->rc) set_error(ticket->err);
    return ticket->rc;
}

}  // namespace fdx
