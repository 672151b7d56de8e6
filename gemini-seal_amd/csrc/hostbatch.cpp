// hostbatch.cpp -- batches of separately allocated HOST ciphertexts through the device.
//
// The reference's objects are separately allocated buffers: a std::vector<seal::Ciphertext> is one IntArray per
// ciphertext (native/src/seal/ciphertext.h:709-721), each handed out by a memory pool. The device entry points of
// include/sealhip.h take one contiguous batch; the *_host entries take what the reference has -- an array of pointers,
// one per ciphertext -- and pipeline it through the GPU in chunks:
//
//     gather (host threads, user buffers -> pinned staging)        chunk c+1
//     host -> device copy              (stream h2d)                chunk c
//     the operation                    (the calling thread's lane) chunk c
//     device -> host copy              (stream d2h)                chunk c
//     scatter (host threads, pinned staging -> user buffers)       chunk c-1
//
// with two staging slots, HIP events between the stages and no host synchronisation other than "the results of chunk
// c-1 have landed". PCIe, not the engine, is what bounds this path (7.3 MB in + 3.7 MB out per
// config-3 multiply+relinearize); the pipeline keeps the link busy in both directions while the kernels run.
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <thread>

#include "engine.hpp"

namespace sealhip
{
    struct HostStage
    {
        int device = -1;
        hipStream_t s_h2d = nullptr, s_d2h = nullptr;
        struct Slot
        {
            char *h_in = nullptr, *h_out = nullptr, *d_in = nullptr, *d_out = nullptr, *d_tmp = nullptr;
            std::size_t cap_in = 0, cap_out = 0, cap_tmp = 0;
            hipEvent_t in_ready = nullptr, done = nullptr, out_ready = nullptr;
        } slot[2];

        void grow(char *&host, char *&dev, std::size_t &cap, std::size_t need, bool with_host)
        {
            if (need <= cap)
                return;
            // every pointer is cleared before the call that may throw: a failed free or allocation must not leave the slot
            // with a dangling pointer and a stale capacity (the next call would reuse it, the destructor free it twice)
            cap = 0;
            if (char *h = host)
            {
                host = nullptr;
                SEALHIP_CHECK(hipHostFree(h));
            }
            if (char *d = dev)
            {
                dev = nullptr;
                SEALHIP_CHECK(hipFree(d));
            }
            if (with_host)
                SEALHIP_CHECK(hipHostMalloc(reinterpret_cast<void **>(&host), need, hipHostMallocDefault));
            SEALHIP_CHECK(hipMalloc(reinterpret_cast<void **>(&dev), need));
            cap = need;
        }
        void ensure(std::size_t in_bytes, std::size_t out_bytes, std::size_t tmp_bytes)
        {
            SEALHIP_CHECK(hipSetDevice(device));
            if (!s_h2d)
            {
                SEALHIP_CHECK(hipStreamCreateWithFlags(&s_h2d, hipStreamNonBlocking));
                SEALHIP_CHECK(hipStreamCreateWithFlags(&s_d2h, hipStreamNonBlocking));
                for (Slot &s : slot)
                {
                    SEALHIP_CHECK(hipEventCreateWithFlags(&s.in_ready, hipEventDisableTiming));
                    SEALHIP_CHECK(hipEventCreateWithFlags(&s.done, hipEventDisableTiming));
                    SEALHIP_CHECK(hipEventCreateWithFlags(&s.out_ready, hipEventDisableTiming));
                }
            }
            for (Slot &s : slot)
            {
                char *none = nullptr;
                grow(s.h_in, s.d_in, s.cap_in, in_bytes, true);
                grow(s.h_out, s.d_out, s.cap_out, out_bytes, true);
                grow(none, s.d_tmp, s.cap_tmp, tmp_bytes, false);
            }
        }
        ~HostStage()
        {
            if (device < 0)
                return;
            (void)hipSetDevice(device);
            if (s_h2d)
                (void)hipStreamSynchronize(s_h2d);
            if (s_d2h)
                (void)hipStreamSynchronize(s_d2h);
            for (Slot &s : slot)
            {
                if (s.h_in)
                    (void)hipHostFree(s.h_in);
                if (s.h_out)
                    (void)hipHostFree(s.h_out);
                if (s.d_in)
                    (void)hipFree(s.d_in);
                if (s.d_out)
                    (void)hipFree(s.d_out);
                if (s.d_tmp)
                    (void)hipFree(s.d_tmp);
                if (s.in_ready)
                    (void)hipEventDestroy(s.in_ready);
                if (s.done)
                    (void)hipEventDestroy(s.done);
                if (s.out_ready)
                    (void)hipEventDestroy(s.out_ready);
            }
            if (s_h2d)
                (void)hipStreamDestroy(s_h2d);
            if (s_d2h)
                (void)hipStreamDestroy(s_d2h);
        }
    };

    void free_host_stage(HostStage *s)
    {
        delete s;
    }

    namespace
    {
        std::size_t host_chunk_items()
        {
            static const std::size_t items = [] {
                const char *env = std::getenv("SEALHIP_HOST_CHUNK");
                const std::size_t v = env ? static_cast<std::size_t>(std::strtoull(env, nullptr, 10)) : 64;
                return v ? v : std::size_t(64);
            }();
            return items;
        }

        std::size_t host_threads()
        {
            static const std::size_t n = [] {
                const char *env = std::getenv("SEALHIP_HOST_THREADS");
                const std::size_t v = env ? static_cast<std::size_t>(std::strtoull(env, nullptr, 10)) : 12;
                return v ? v : std::size_t(12);
            }();
            return n;
        }

        // items [0, m) split over a few host threads (the copies between user buffers and the pinned staging area are
        // what a single core cannot keep up with: one core moves ~10 GB/s, the link takes ~50)
        template <class F>
        void parallel_items(std::size_t m, F &&body)
        {
            const unsigned hw = std::thread::hardware_concurrency();
            const std::size_t nthreads = std::max<std::size_t>(1, std::min<std::size_t>({ host_threads(), hw ? hw : 1, m }));
            if (nthreads == 1)
            {
                for (std::size_t i = 0; i < m; i++)
                    body(i);
                return;
            }
            // joins whatever was started, also when a thread cannot be created or the body throws on this thread: a joinable
            // std::thread that is destroyed calls std::terminate -- inside a C ABI entry point
            struct Joiner
            {
                std::vector<std::thread> pool;
                ~Joiner()
                {
                    for (auto &th : pool)
                        if (th.joinable())
                            th.join();
                }
            } workers;
            workers.pool.reserve(nthreads);
            const auto stripe = [&](std::size_t t) {
                for (std::size_t i = t; i < m; i += nthreads)
                    body(i);
            };
            for (std::size_t t = 1; t < nthreads; t++)
                workers.pool.emplace_back(stripe, t);
            stripe(0); // the calling thread takes a stripe itself
        }
    } // namespace

    void run_host_batch(Engine &e, const HostBatchIO &io, std::size_t count, const HostChunkFn &fn)
    {
        if (count == 0)
            return;
        for (const auto &a : io.in)
            for (std::size_t i = 0; i < count; i++)
                if (!a.ptrs[i])
                    throw std::invalid_argument("null ciphertext pointer in the batch");
        for (const auto &a : io.out)
            for (std::size_t i = 0; i < count; i++)
                if (!a.ptrs[i])
                    throw std::invalid_argument("null ciphertext pointer in the batch");
        Lane &lane = e.lane();
        if (lane.capturing)
            throw std::logic_error("host batches synchronise: they cannot be captured into a graph");
        if (!lane.stage)
        {
            lane.stage = new HostStage();
            lane.stage->device = e.device;
        }
        HostStage &st = *lane.stage;
        const std::size_t chunk = std::min(host_chunk_items(), count);
        std::size_t in_item = 0, out_item = 0;
        for (const auto &a : io.in)
            in_item += a.words * sizeof(u64);
        for (const auto &a : io.out)
            out_item += a.words * sizeof(u64);
        st.ensure(std::max<std::size_t>(chunk * in_item, 256), std::max<std::size_t>(chunk * out_item, 256),
                  std::max<std::size_t>(chunk * io.tmp_words * sizeof(u64), 256));

        const std::size_t nchunks = (count + chunk - 1) / chunk;
        auto scatter = [&](std::size_t c) {
            HostStage::Slot &s = st.slot[c & 1];
            SEALHIP_CHECK(hipEventSynchronize(s.out_ready));
            e.check_fault(); // results are about to become host-visible
            const std::size_t off = c * chunk, m = std::min(chunk, count - off);
            std::size_t base = 0;
            for (const auto &a : io.out)
            {
                const char *src = s.h_out + base;
                parallel_items(m, [&](std::size_t i) {
                    std::memcpy(a.ptrs[off + i], src + i * a.words * sizeof(u64), a.words * sizeof(u64));
                });
                base += chunk * a.words * sizeof(u64);
            }
        };
        try
        {
            for (std::size_t c = 0; c < nchunks; c++)
            {
                HostStage::Slot &s = st.slot[c & 1];
                const std::size_t off = c * chunk, m = std::min(chunk, count - off);
                // (this slot is free: the scatter of chunk c-2, one iteration ago, waited for everything that used it)
                std::size_t base = 0;
                std::vector<u64 *> d_in, d_out;
                for (const auto &a : io.in)
                {
                    char *dst = s.h_in + base;
                    parallel_items(m, [&](std::size_t i) {
                        std::memcpy(dst + i * a.words * sizeof(u64), a.ptrs[off + i], a.words * sizeof(u64));
                    });
                    SEALHIP_CHECK(hipMemcpyAsync(s.d_in + base, dst, m * a.words * sizeof(u64), hipMemcpyHostToDevice, st.s_h2d));
                    d_in.push_back(reinterpret_cast<u64 *>(s.d_in + base));
                    base += chunk * a.words * sizeof(u64);
                }
                SEALHIP_CHECK(hipEventRecord(s.in_ready, st.s_h2d));
                SEALHIP_CHECK(hipStreamWaitEvent(lane.stream, s.in_ready, 0));
                base = 0;
                for (const auto &a : io.out)
                {
                    d_out.push_back(reinterpret_cast<u64 *>(s.d_out + base));
                    base += chunk * a.words * sizeof(u64);
                }
                fn(e, d_in, d_out, reinterpret_cast<u64 *>(s.d_tmp), m);
                SEALHIP_CHECK(hipEventRecord(s.done, lane.stream));
                SEALHIP_CHECK(hipStreamWaitEvent(st.s_d2h, s.done, 0));
                base = 0;
                for (const auto &a : io.out)
                {
                    SEALHIP_CHECK(hipMemcpyAsync(s.h_out + base, s.d_out + base, m * a.words * sizeof(u64), hipMemcpyDeviceToHost,
                                                 st.s_d2h));
                    base += chunk * a.words * sizeof(u64);
                }
                SEALHIP_CHECK(hipEventRecord(s.out_ready, st.s_d2h));
                if (c > 0)
                    scatter(c - 1); // overlaps the device work of chunk c enqueued above
            }
            scatter(nchunks - 1);
        }
        catch (...)
        {
            // drain before the staging buffers can be reused or the caller's buffers go away
            (void)hipStreamSynchronize(st.s_h2d);
            (void)hipStreamSynchronize(lane.stream);
            (void)hipStreamSynchronize(st.s_d2h);
            throw;
        }
    }
} // namespace sealhip
