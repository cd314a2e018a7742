// host_pool.hpp -- the persistent worker threads of the host-side helpers (magphase_host.cpp, magphase_plan.cpp).
#pragma once
#include <unistd.h>

#include <atomic>
#include <condition_variable>
#include <exception>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

namespace mpx_host {

// A few persistent worker threads per CALLING thread (thread_local: the reader, compute and writer threads of iobatch each
// get their own, so their calls still overlap).  Spawning std::threads per call cost ~0.25 ms for eight -- half of the
// time of one 16 MB conversion chunk of the array API (chunks of 8 / 16 / 32 / 64 MB: 22 / 16 / 11 / 11.5 ms per call).
class WorkerPool {
   public:
    ~WorkerPool() { shutdown(); }

    // body() on the calling thread and on n_workers pool threads; returns when all of them are done with it
    void run(int n_workers, const std::function<void()>& body) {
        if (pid_ != getpid()) {   // forked child: the parent's workers do not exist here (their handles are abandoned)
            abandon();
            pid_ = getpid();
        }
        {
            std::unique_lock<std::mutex> lk(mu_);
            while ((int)th_.size() < n_workers) {
                const int idx = (int)th_.size();
                th_.emplace_back([this, idx] { loop(idx); });
            }
            body_ = &body;
            want_ = n_workers;
            finished_ = 0;
            worker_err_ = nullptr;
            ++gen_;
        }
        cv_work_.notify_all();
        // `body` lives on the caller's stack and the workers dereference it: if it throws on this thread (bad_alloc in a
        // file reader), the workers are waited for BEFORE the exception leaves this frame
        std::exception_ptr err;
        try {
            body();
        } catch (...) {
            err = std::current_exception();
        }
        {
            std::unique_lock<std::mutex> lk(mu_);
            cv_done_.wait(lk, [&] { return finished_ == want_; });
            body_ = nullptr;
            // a worker's copy of the body ran OTHER indices than this thread's: its exception (bad_alloc in a file
            // reader) is not reproduced here, so the first one is carried over and rethrown
            if (!err) err = worker_err_;
            worker_err_ = nullptr;
        }
        if (err) std::rethrow_exception(err);
    }

   private:
    void loop(int idx) {
        unsigned seen = 0;
        for (;;) {
            const std::function<void()>* body = nullptr;
            {
                std::unique_lock<std::mutex> lk(mu_);
                cv_work_.wait(lk, [&] { return stop_ || gen_ != seen; });
                if (stop_) return;
                seen = gen_;
                if (idx < want_) body = body_;
            }
            if (body) {
                std::exception_ptr err;
                try {
                    (*body)();
                } catch (...) {   // must not terminate the process from a pool thread: handed to run(), which rethrows it
                    err = std::current_exception();
                }
                std::unique_lock<std::mutex> lk(mu_);
                if (err && !worker_err_) worker_err_ = err;
                if (++finished_ == want_) cv_done_.notify_one();
            }
        }
    }
    void abandon() {   // handles of threads that do not exist in this process: neither joined nor destroyed
        if (!th_.empty()) (void)new std::vector<std::thread>(std::move(th_));
        th_.clear();
    }
    void shutdown() {
        if (pid_ != getpid()) {
            abandon();
            return;
        }
        {
            std::unique_lock<std::mutex> lk(mu_);
            stop_ = true;
        }
        cv_work_.notify_all();
        for (auto& t : th_) t.join();
        th_.clear();
    }
    std::mutex mu_;
    std::condition_variable cv_work_, cv_done_;
    std::vector<std::thread> th_;
    const std::function<void()>* body_ = nullptr;
    std::exception_ptr worker_err_;
    int want_ = 0, finished_ = 0;
    unsigned gen_ = 0;
    bool stop_ = false;
    pid_t pid_ = getpid();
};

// ONE pool per calling thread, shared by every parallel_for instantiation (a `static thread_local` inside the function
// template gave each of the six call sites its own pool per calling thread: ~126 parked threads instead of ~21).
inline WorkerPool& thread_pool() {
    static thread_local WorkerPool pool;
    return pool;
}

// Runs fn(i) for i in [0, n) on up to n_threads threads (work stealing through one atomic counter).
template <typename F>
void parallel_for(int n, int n_threads, F fn) {
    if (n <= 0) return;
    const int nt = n_threads <= 1 ? 1 : (n_threads < n ? n_threads : n);
    if (nt == 1) {
        for (int i = 0; i < n; ++i) fn(i);
        return;
    }
    std::atomic<int> next(0);
    const std::function<void()> worker = [&] {
        for (int i = next.fetch_add(1); i < n; i = next.fetch_add(1)) fn(i);
    };
    thread_pool().run(nt - 1, worker);
}

}  // namespace mpx_host
