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
//
// Round 4, measured (profiles/r04/pcie_probe.txt, host_batch_probe.txt): the link gives 48.6 GB/s each way when both directions
// run; through the staging copies (12 host threads, the calling thread gathers and scatters every chunk) a config-3 batch
// sees 37-41 GB/s in. A caller whose buffers live in memory it keeps (the blocks of a MemoryPool,
// native/src/seal/util/mempool.cpp:45,145) can pin them in place once -- sealhip_host_register -- and an array whose items all
// lie in registered ranges skips the staging copy: the DMA engine reads / writes the caller's buffers directly (42-45 GB/s
// in, +10-20 % ciphertexts per second), the host threads have nothing to do for it.
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <shared_mutex>
#include <thread>

#include "engine.hpp"

namespace sealhip
{
    // ---- host ranges pinned in place (process-wide: a registration is visible to every device, hipHostRegisterPortable)
    namespace
    {
        std::shared_mutex g_reg_mutex;
        std::map<std::uintptr_t, std::size_t> g_registered; // start -> bytes, disjoint

        bool range_registered(const void *p, std::size_t bytes)
        {
            const std::uintptr_t a = reinterpret_cast<std::uintptr_t>(p);
            auto it = g_registered.upper_bound(a);
            if (it == g_registered.begin())
                return false;
            --it;
            return a >= it->first && bytes <= it->second && a - it->first <= it->second - bytes;
        }
    } // namespace

    void host_register(const Engine &e, void *ptr, std::size_t bytes)
    {
        if (!ptr || !bytes)
            throw std::invalid_argument("host_register: empty range");
        if (e.device < 0)
            throw std::logic_error("host_register needs a device context");
        const std::uintptr_t a = reinterpret_cast<std::uintptr_t>(ptr);
        std::unique_lock<std::shared_mutex> lock(g_reg_mutex);
        // overlap with a range already pinned: the runtime would refuse it; say which argument is wrong
        auto it = g_registered.upper_bound(a);
        if (it != g_registered.end() && it->first - a < bytes)
            throw std::invalid_argument("host_register: the range overlaps a registered one");
        if (it != g_registered.begin())
        {
            --it;
            if (a - it->first < it->second)
                throw std::invalid_argument("host_register: the range overlaps a registered one");
        }
        SEALHIP_CHECK(hipSetDevice(e.device));
        SEALHIP_CHECK(hipHostRegister(ptr, bytes, hipHostRegisterPortable));
        g_registered.emplace(a, bytes);
    }

    void host_unregister(const Engine &e, void *ptr)
    {
        std::unique_lock<std::shared_mutex> lock(g_reg_mutex);
        const auto it = g_registered.find(reinterpret_cast<std::uintptr_t>(ptr));
        if (it == g_registered.end())
            throw std::invalid_argument("host_unregister: not the start of a registered range");
        if (e.device >= 0)
            SEALHIP_CHECK(hipSetDevice(e.device));
        // (a transfer still in flight on the range is the caller's to wait for: the *_host entries return synchronised)
        SEALHIP_CHECK(hipHostUnregister(ptr));
        g_registered.erase(it);
    }

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
        // an array whose items of this chunk all lie inside registered ranges is copied by the DMA engine from / to the
        // caller's buffers themselves (the registry is read under its lock: a concurrent unregister of a range in use is
        // the caller's error, as freeing the buffer would be)
        const auto all_registered = [&](auto const &a, std::size_t off, std::size_t m) {
            std::shared_lock<std::shared_mutex> lock(g_reg_mutex);
            if (g_registered.empty())
                return false;
            for (std::size_t i = 0; i < m; i++)
                if (!range_registered(a.ptrs[off + i], a.words * sizeof(u64)))
                    return false;
            return true;
        };
        std::vector<char> out_direct(nchunks * io.out.size(), 0);
        auto scatter = [&](std::size_t c) {
            HostStage::Slot &s = st.slot[c & 1];
            SEALHIP_CHECK(hipEventSynchronize(s.out_ready));
            e.check_fault(); // results are about to become host-visible
            const std::size_t off = c * chunk, m = std::min(chunk, count - off);
            std::size_t base = 0, ai = 0;
            for (const auto &a : io.out)
            {
                const char *src = s.h_out + base;
                if (!out_direct[c * io.out.size() + ai]) // (else: already in the caller's buffers)
                    parallel_items(m, [&](std::size_t i) {
                        std::memcpy(a.ptrs[off + i], src + i * a.words * sizeof(u64), a.words * sizeof(u64));
                    });
                base += chunk * a.words * sizeof(u64);
                ai++;
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
                    const std::size_t item_bytes = a.words * sizeof(u64);
                    if (all_registered(a, off, m))
                    {
                        for (std::size_t i = 0; i < m; i++)
                            SEALHIP_CHECK(hipMemcpyAsync(s.d_in + base + i * item_bytes, a.ptrs[off + i], item_bytes,
                                                         hipMemcpyHostToDevice, st.s_h2d));
                    }
                    else
                    {
                        parallel_items(m, [&](std::size_t i) { std::memcpy(dst + i * item_bytes, a.ptrs[off + i], item_bytes); });
                        SEALHIP_CHECK(hipMemcpyAsync(s.d_in + base, dst, m * item_bytes, hipMemcpyHostToDevice, st.s_h2d));
                    }
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
                std::size_t ai = 0;
                for (const auto &a : io.out)
                {
                    const std::size_t item_bytes = a.words * sizeof(u64);
                    if (all_registered(a, off, m))
                    {
                        out_direct[c * io.out.size() + ai] = 1;
                        for (std::size_t i = 0; i < m; i++)
                            SEALHIP_CHECK(hipMemcpyAsync(a.ptrs[off + i], s.d_out + base + i * item_bytes, item_bytes,
                                                         hipMemcpyDeviceToHost, st.s_d2h));
                    }
                    else
                        SEALHIP_CHECK(hipMemcpyAsync(s.h_out + base, s.d_out + base, m * item_bytes, hipMemcpyDeviceToHost, st.s_d2h));
                    base += chunk * item_bytes;
                    ai++;
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
