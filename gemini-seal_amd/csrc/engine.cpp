// engine.cpp -- context construction: regenerates every table/constant the hot path needs
// (what SEALContext / ContextData / RNSTool / NTTTables hold in the reference:
// native/src/seal/context.cpp:455-540, util/rns.cpp:539-729, util/ntt.cpp:37-99) and uploads it.
#include "engine.hpp"

#include <algorithm>
#include <atomic>
#include <cstdlib>
#include <cstring>

namespace sealhip
{
    namespace
    {
        template <class T>
        T *upload(Engine &e, std::vector<void *> &owned, const T *host, std::size_t count)
        {
            T *dev = nullptr;
            SEALHIP_CHECK(hipMalloc(reinterpret_cast<void **>(&dev), sizeof(T) * (count ? count : 1)));
            owned.push_back(dev);
            if (count)
                SEALHIP_CHECK(hipMemcpy(dev, host, sizeof(T) * count, hipMemcpyHostToDevice));
            (void)e;
            return dev;
        }

        RowMap make_map(const std::vector<unsigned short> &ids)
        {
            RowMap m{};
            if (ids.size() > static_cast<std::size_t>(kMaxRows))
                throw std::invalid_argument("too many rows per polynomial");
            m.rows = static_cast<int>(ids.size());
            for (std::size_t i = 0; i < ids.size(); i++)
                m.prime[i] = ids[i];
            return m;
        }
    } // namespace

    unsigned long long next_pool_id();

    std::unique_ptr<Engine> make_engine(int scheme, int logn, const u64 *key_moduli, int n_key, int nsp, u64 t,
                                        bool strict, int device)
    {
        if (scheme != 1 && scheme != 2)
            throw std::invalid_argument("unsupported scheme"); // evaluator.cpp:262-263
        if (logn < 3 || logn > 16)
            // SEAL_POLY_MOD_DEGREE_MAX = 65536 (util/defines.h:53); the fork's inverse NTT is only defined for
            // N >= 8 (ntt.cpp:352-402)
            throw std::invalid_argument("poly_modulus_degree is invalid");
        if (n_key < 2 || n_key > kMaxModuli || nsp < 1 || n_key <= nsp)
            throw std::invalid_argument("coeff_modulus / n_special_primes invalid"); // context.cpp:527-530
        if (scheme == 1 && t < 2)
            throw std::invalid_argument("plain_modulus is invalid");
        auto e = std::make_unique<Engine>();
        e->scheme = scheme;
        e->logn = logn;
        e->n = std::size_t(1) << logn;
        e->n_key = n_key;
        e->nsp = nsp;
        e->k_first = n_key - nsp;
        e->t = scheme == 1 ? t : 0;
        e->mode_strict = strict;
        e->use_half_kernel = true;
        e->device = device;
        e->key_moduli.assign(key_moduli, key_moduli + n_key);
        for (int i = 0; i < n_key; i++)
        {
            HostModulus check(e->key_moduli[i]); // range check
            (void)check;
            for (int j = 0; j < i; j++)
                if (e->key_moduli[i] == e->key_moduli[j])
                    throw std::invalid_argument("coeff_modulus primes must be distinct");
        }
        // auxiliary 60-bit primes for BEHZ (rns.cpp:587): m_sk, gamma, B_0, B_1, ...
        if (scheme == 1)
            e->aux_primes = get_primes(e->n, 60, static_cast<std::size_t>(n_key) + 3);
        // BFV with a prime plain modulus = 1 (mod 2N): the plain NTT tables BatchEncoder uses (context.cpp:262-275)
        const bool batching = scheme == 1 && (t - 1) % (2 * static_cast<u64>(e->n)) == 0 && is_prime_u64(t);
        const int n_primes = n_key + static_cast<int>(e->aux_primes.size()) + (batching ? 1 : 0);
        if (batching)
            e->plain_prime = n_primes - 1;
        e->tables.resize(n_primes);
        for (int i = 0; i < n_primes; i++)
        {
            const u64 p = i == e->plain_prime ? t : i < n_key ? e->key_moduli[i] : e->aux_primes[i - n_key];
            if (i == n_key + 1 && i != e->plain_prime)
            {
                // gamma is never transformed; keep only its modulus
                e->tables[i].logn = logn;
                e->tables[i].p = p;
                continue;
            }
            e->tables[i].build(logn, p);
        }
        if (device < 0)
            return e;

        SEALHIP_CHECK(hipSetDevice(device));
        e->lanes = std::make_shared<LanePool>();
        e->lanes->device = device;
        e->lanes->id = next_pool_id();
        // sticky device-side failure flag in host-mapped (coherent) memory: no copy is needed to read it after a sync
        SEALHIP_CHECK(ntt_init_kernels());
        std::vector<PrimeDev> pd(n_primes);
        for (int i = 0; i < n_primes; i++)
        {
            const HostNttTables &tb = e->tables[i];
            HostModulus m(tb.p);
            PrimeDev &d = pd[i];
            d.p = tb.p;
            d.two_p = tb.p << 1;
            d.rdp = tb.rdp ? tb.rdp : shoup(1, tb.p);
            d.cr0 = m.cr0;
            d.cr1 = m.cr1;
            d.inv_n = tb.inv_n;
            d.inv_n_shoup = tb.inv_n_shoup;
            d.inv_n_w = tb.inv_n_w;
            d.inv_n_w_shoup = tb.inv_n_w_shoup;
            {
                // -p^{-1} mod 2^64 by Newton iteration (p odd; the even "modulus" 2^32 never uses REDC)
                u64 inv = tb.p;
                for (int it = 0; it < 6; it++)
                    inv *= 2 - tb.p * inv;
                d.ninv = 0 - inv;
            }
            d.fwd = tb.fwd.empty() ? nullptr : upload<u64>(*e, e->owned, tb.fwd.data(), tb.fwd.size());
            d.inv = tb.inv.empty() ? nullptr : upload<u64>(*e, e->owned, tb.inv.data(), tb.inv.size());
            d.fwd_d = d.inv_d = nullptr;
            d.p_d = static_cast<double>(tb.p);
            d.pinv_d = 1.0 / d.p_d;
            if (tb.p < kFpPrimeBound && tb.logn >= 14 && !tb.fwd.empty())
            {
                // the single-pass kernels' floating-point variant (ntt.hip): the twiddles w, without the Shoup quotients
                std::vector<double> f(tb.fwd.size() / 2), v(tb.inv.size() / 2);
                for (std::size_t j = 0; j < f.size(); j++)
                    f[j] = static_cast<double>(tb.fwd[2 * j]);
                for (std::size_t j = 0; j < v.size(); j++)
                    v[j] = static_cast<double>(tb.inv[2 * j]);
                d.fwd_d = upload<double>(*e, e->owned, f.data(), f.size());
                d.inv_d = upload<double>(*e, e->owned, v.data(), v.size());
            }
        }
        e->d_primes = upload<PrimeDev>(*e, e->owned, pd.data(), pd.size());
        return e;
    }

    // ---------------------------------------------------------------- lanes (per-host-thread execution state)
    Lane::~Lane()
    {
        if (device < 0)
            return;
        (void)hipSetDevice(device);
        if (stream)
            (void)hipStreamSynchronize(stream);
        for (ProfRecord &r : prof)
        {
            if (r.start)
                (void)hipEventDestroy(r.start);
            if (r.stop)
                (void)hipEventDestroy(r.stop);
        }
        if (stage)
            free_host_stage(stage);
        if (ws)
            (void)hipFree(ws);
        if (d_tickets)
            (void)hipFree(d_tickets);
        if (h_fault)
            (void)hipHostFree(h_fault);
        if (own_stream && stream)
            (void)hipStreamDestroy(stream);
    }

    Lane *LanePool::take()
    {
        std::lock_guard<std::mutex> lock(mu);
        if (!idle.empty())
        {
            Lane *l = idle.back();
            idle.pop_back();
            l->tsink = l->tsink_cur = l->tsink_arm = nullptr; // (give() cleared them; a lane is handed out without a sink)
            l->tsink_cap = l->tsink_base = 0;
            return l;
        }
        auto l = std::make_unique<Lane>();
        l->device = device;
        SEALHIP_CHECK(hipSetDevice(device));
        SEALHIP_CHECK(hipStreamCreateWithFlags(&l->stream, hipStreamNonBlocking));
        l->own_stream = true;
        SEALHIP_CHECK(hipHostMalloc(reinterpret_cast<void **>(&l->h_fault), sizeof(unsigned), hipHostMallocMapped));
        *l->h_fault = 0;
        SEALHIP_CHECK(hipHostGetDevicePointer(reinterpret_cast<void **>(&l->d_fault), l->h_fault, 0));
        all.push_back(std::move(l));
        return all.back().get();
    }

    void LanePool::give(Lane *lane)
    {
        // The lane outlives the thread that held it, the transparency sink does not: sealhip_transparency_sink registers a
        // raw device pointer of THAT thread (include/sealhip.h: "the calling thread's lane"). A thread that takes the lane
        // later must not inherit it -- its first Evaluator call would clear and write flags in a buffer it never
        // registered, possibly freed. No HIP call here: this runs from a thread_local destructor, possibly
        // while the process is shutting down. A pending device fault stays sticky on purpose: work of the exiting thread
        // may still be in flight on the lane's stream, and sealhip_synchronize (all lanes) is where it surfaces.
        lane->tsink = lane->tsink_cur = lane->tsink_arm = nullptr;
        lane->tsink_cap = lane->tsink_base = 0;
        std::lock_guard<std::mutex> lock(mu);
        idle.push_back(lane);
    }

    namespace
    {
        // the lanes this thread holds, one per context it has called into; returned to their pools when the thread exits
        struct ThreadLanes
        {
            struct Held
            {
                std::weak_ptr<LanePool> pool;
                Lane *lane;
            };
            std::vector<Held> held;
            ~ThreadLanes()
            {
                for (Held &h : held)
                    if (auto p = h.pool.lock())
                        p->give(h.lane);
            }
        };
        thread_local ThreadLanes t_lanes;
        // one-entry cache of the last lookup, keyed by the pool's serial number (never reused, unlike its address)
        thread_local unsigned long long t_cur_pool_id = 0;
        thread_local Lane *t_cur_lane = nullptr;
        std::atomic<unsigned long long> g_pool_serial{ 0 };
    } // namespace

    unsigned long long next_pool_id()
    {
        return ++g_pool_serial;
    }

    Lane &Engine::lane() const
    {
        if (!lanes)
            throw std::logic_error("host-only context: no device");
        if (t_cur_pool_id == lanes->id)
            return *t_cur_lane;
        auto &held = t_lanes.held;
        for (std::size_t i = 0; i < held.size();)
        {
            auto p = held[i].pool.lock();
            if (!p)
            {
                held.erase(held.begin() + static_cast<std::ptrdiff_t>(i)); // the context is gone (and its lanes with it)
                continue;
            }
            if (p.get() == lanes.get())
            {
                t_cur_pool_id = lanes->id;
                t_cur_lane = held[i].lane;
                return *t_cur_lane;
            }
            i++;
        }
        Lane *l = lanes->take();
        held.push_back({ lanes, l });
        t_cur_pool_id = lanes->id;
        t_cur_lane = l;
        return *l;
    }

    void Engine::sync_and_check(bool all_lanes) const
    {
        if (all_lanes)
        {
            std::vector<hipStream_t> streams;
            {
                std::lock_guard<std::mutex> lock(lanes->mu);
                for (auto &l : lanes->all)
                    if (!l->capturing)
                        streams.push_back(l->stream);
            }
            for (hipStream_t s : streams)
                SEALHIP_CHECK(hipStreamSynchronize(s));
            // every lane has been waited for: a failure of any of them fails this call (and is consumed by it)
            bool any = false;
            {
                std::lock_guard<std::mutex> lock(lanes->mu);
                for (auto &l : lanes->all)
                    if (l->h_fault && __atomic_exchange_n(l->h_fault, 0u, __ATOMIC_ACQ_REL))
                        any = true;
            }
            if (any)
                throw std::runtime_error("forward NTT: sibling workgroup wait timed out; results of that launch are invalid");
            return;
        }
        SEALHIP_CHECK(hipStreamSynchronize(lane().stream));
        check_fault();
    }

    void Engine::check_fault() const
    {
        Lane &l = lane();
        if (l.h_fault && __atomic_exchange_n(l.h_fault, 0u, __ATOMIC_ACQ_REL))
            throw std::runtime_error("forward NTT: sibling workgroup wait timed out; results of that launch are invalid");
    }

    unsigned *Engine::ntt_tickets(std::size_t nrows) const
    {
        Lane &l = lane();
        if (nrows > l.tickets_cap)
        {
            if (l.capturing)
                return nullptr; // reported by the launcher; run the sequence once before capturing
            if (l.d_tickets)
            {
                if (hipStreamSynchronize(l.stream) != hipSuccess || hipFree(l.d_tickets) != hipSuccess)
                    return nullptr;
                l.d_tickets = nullptr;
                l.tickets_cap = 0;
            }
            std::size_t cap = 1;
            while (cap < nrows)
                cap <<= 1;
            l.alloc_generation++;
            if (hipMalloc(reinterpret_cast<void **>(&l.d_tickets), cap * sizeof(unsigned)) != hipSuccess)
                return nullptr;
            l.tickets_cap = cap;
        }
        if (hipMemsetAsync(l.d_tickets, 0, nrows * sizeof(unsigned), l.stream) != hipSuccess)
            return nullptr;
        return l.d_tickets;
    }

    void Engine::prof_begin(const char *tag, double units) const
    {
        Lane &l = lane();
        ProfRecord r{ tag, nullptr, nullptr, units };
        if (hipEventCreate(&r.start) != hipSuccess || hipEventCreate(&r.stop) != hipSuccess)
            return;
        (void)hipEventRecord(r.start, l.stream);
        l.prof.push_back(r);
    }

    void Engine::prof_end() const
    {
        Lane &l = lane();
        if (!l.prof.empty() && l.prof.back().stop)
            (void)hipEventRecord(l.prof.back().stop, l.stream);
    }

    Engine::~Engine()
    {
        if (device < 0)
            return;
        (void)hipSetDevice(device);
        if (lanes)
        {
            // every lane drains its stream and frees its arena; threads that still hold one see an expired pool
            std::lock_guard<std::mutex> lock(lanes->mu);
            lanes->all.clear();
            lanes->idle.clear();
        }
        lanes.reset();
        for (auto &kv : levels)
            for (void *p : kv.second->owned)
                (void)hipFree(p);
        for (auto &kv : galois_tables)
            (void)hipFree(kv.second);
        for (void *p : owned)
            (void)hipFree(p);
        if (d_batch_map)
            (void)hipFree(d_batch_map);
        if (d_ckks_map)
            (void)hipFree(d_ckks_map);
        if (d_ckks_roots)
            (void)hipFree(d_ckks_roots);
        if (d_ckks_inv_roots)
            (void)hipFree(d_ckks_inv_roots);
        for (auto &kv : ckks_decode)
            (void)hipFree(kv.second);
    }

    LevelTools &Engine::level_host(int k)
    {
        if (k < 1 || k > n_key)
            throw std::invalid_argument("level k out of range");
        std::lock_guard<std::mutex> lock(mu);
        auto it = levels.find(k);
        if (it != levels.end())
            return *it->second;
        auto lt = std::make_unique<LevelTools>();
        std::vector<u64> q(key_moduli.begin(), key_moduli.begin() + k);
        lt->host_rns = std::make_unique<HostRnsTool>();
        if (scheme == 1)
            lt->host_rns->build(n, q, t, aux_primes);
        else
        {
            // CKKS: only inv_q_last_mod_q is used on the path (rns.cpp:719-728)
            lt->host_rns->n = n;
            lt->host_rns->q = q;
            lt->host_rns->inv_q_last_mod_q.assign(k - 1, 0);
            for (int i = 0; i + 1 < k; i++)
                if (!invmod(q[k - 1], q[i], lt->host_rns->inv_q_last_mod_q[i]))
                    throw std::logic_error("invalid rns bases");
        }
        // row maps
        std::vector<unsigned short> ids_q, ids_bsk, ids_key, ids_qbsk;
        for (int i = 0; i < k; i++)
            ids_q.push_back(static_cast<unsigned short>(i));
        ids_key = ids_q;
        if (k <= k_first)
            for (int j = 0; j < nsp; j++)
                ids_key.push_back(static_cast<unsigned short>(k_first + j));
        if (scheme == 1)
        {
            const std::size_t B = lt->host_rns->B_size;
            for (std::size_t j = 0; j < B; j++)
                ids_bsk.push_back(static_cast<unsigned short>(n_key + 2 + j));
            ids_bsk.push_back(static_cast<unsigned short>(n_key)); // m_sk last (rns.cpp:600)
        }
        ids_qbsk = ids_q;
        ids_qbsk.insert(ids_qbsk.end(), ids_bsk.begin(), ids_bsk.end());
        lt->map_q = make_map(ids_q);
        lt->map_bsk = make_map(ids_bsk);
        lt->map_key = make_map(ids_key);
        lt->map_qbsk = make_map(ids_qbsk);
        auto &ref = *lt;
        levels.emplace(k, std::move(lt));
        return ref;
    }

    LevelTools &Engine::level(int k)
    {
        LevelTools &lt = level_host(k);
        if (device < 0)
            throw std::logic_error("host-only context: no device");
        std::lock_guard<std::mutex> lock(mu);
        if (lt.d_rns)
            return lt;
        if (lane().capturing) // the uploads below are synchronous copies: they would invalidate a relaxed capture
            throw std::logic_error("the constants of this level are not built yet: run the sequence once before capturing");
        SEALHIP_CHECK(hipSetDevice(device));
        const HostRnsTool &hr = *lt.host_rns;
        RnsDev rd{};
        rd.k = k;
        rd.t = t;
        for (int i = 0; i < k; i++)
            rd.q_prime[i] = static_cast<unsigned>(i);
        for (int i = 0; i + 1 < k; i++)
            rd.inv_q_last_mod_q[i] = hr.inv_q_last_mod_q[i];
        if (scheme == 1)
        {
            const int nB = static_cast<int>(hr.Bsk.size()), B = static_cast<int>(hr.B_size);
            rd.nB = nB;
            rd.B = B;
            for (int i = 0; i < k; i++)
            {
                rd.q_inv[i] = hr.q_to_Bsk.inv_punct[i];
                rd.q_mt_inv[i] = mulmod(hr.m_tilde % hr.q[i], hr.q_to_Bsk.inv_punct[i], hr.q[i]);
                rd.q_to_mt[i] = hr.q_to_m_tilde.matrix[i];
                rd.prod_B_mod_q[i] = hr.prod_B_mod_q[i];
            }
            rd.q_to_Bsk = upload<u64>(*this, lt.owned, hr.q_to_Bsk.matrix.data(), hr.q_to_Bsk.matrix.size());
            rd.inv_prod_q_mod_mt = hr.inv_prod_q_mod_m_tilde;
            for (int j = 0; j < nB; j++)
            {
                rd.prod_q_mod_Bsk[j] = hr.prod_q_mod_Bsk[j];
                rd.inv_prod_q_mod_Bsk[j] = hr.inv_prod_q_mod_Bsk[j];
                rd.inv_mt_mod_Bsk[j] = hr.inv_m_tilde_mod_Bsk[j];
                rd.bsk_prime[j] = lt.map_bsk.prime[j];
            }
            for (int i = 0; i < B; i++)
            {
                rd.B_inv[i] = hr.B_to_q.inv_punct[i];
                rd.B_to_msk[i] = hr.B_to_m_sk.matrix[i];
            }
            rd.B_to_q = upload<u64>(*this, lt.owned, hr.B_to_q.matrix.data(), hr.B_to_q.matrix.size());
            rd.inv_prod_B_mod_msk = hr.inv_prod_B_mod_m_sk;
            // folded constants of the fused kernels
            std::vector<u64> L1(static_cast<std::size_t>(nB) * k), G2(static_cast<std::size_t>(nB) * k);
            for (int j = 0; j < nB; j++)
            {
                const u64 b = hr.Bsk[j];
                const u64 inv_mt = hr.inv_m_tilde_mod_Bsk[j];
                u64 g = hr.inv_prod_q_mod_Bsk[j];
                if (j < B)
                    g = mulmod(g, hr.B_to_q.inv_punct[j], b);
                rd.lift_L2[j] = mulmod(hr.prod_q_mod_Bsk[j], inv_mt, b);
                rd.floor_G1[j] = mulmod(t % b, g, b);
                for (int i = 0; i < k; i++)
                {
                    const u64 m = hr.q_to_Bsk.matrix[static_cast<std::size_t>(j) * k + i];
                    L1[static_cast<std::size_t>(j) * k + i] = mulmod(m, inv_mt, b);
                    const u64 mg = mulmod(m, g, b);
                    G2[static_cast<std::size_t>(j) * k + i] = mg ? b - mg : 0;
                }
            }
            for (int i = 0; i < k; i++)
                rd.floor_F0[i] = mulmod(t % hr.q[i], hr.q_to_Bsk.inv_punct[i], hr.q[i]);
            rd.lift_L1 = upload<u64>(*this, lt.owned, L1.data(), L1.size());
            rd.floor_G2 = upload<u64>(*this, lt.owned, G2.data(), G2.size());
            // Montgomery (x 2^64 mod prime) and Shoup companions
            auto mont = [](u64 c, u64 p) { return static_cast<u64>((static_cast<u128>(c) << 64) % p); };
            std::vector<u64> L1m(L1.size()), G2m(G2.size()), BQm(hr.B_to_q.matrix.size());
            u64 max_q = 0, max_b = 0;
            for (int i = 0; i < k; i++)
                max_q = std::max(max_q, hr.q[i]);
            for (int j = 0; j < nB; j++)
            {
                const u64 b = hr.Bsk[j];
                max_b = std::max(max_b, b);
                rd.lift_L2m[j] = mont(rd.lift_L2[j], b);
                rd.floor_G1m[j] = mont(rd.floor_G1[j], b);
                for (int i = 0; i < k; i++)
                {
                    L1m[static_cast<std::size_t>(j) * k + i] = mont(L1[static_cast<std::size_t>(j) * k + i], b);
                    G2m[static_cast<std::size_t>(j) * k + i] = mont(G2[static_cast<std::size_t>(j) * k + i], b);
                }
            }
            for (int j = 0; j < nB; j++)
            {
                // Bsk row j is prime id n_key + (j < B ? 2 + j : 0) (m_sk last); fold its top-layer factors into G1
                const u64 b = hr.Bsk[j];
                const HostNttTables &tb = tables[n_key + (j < B ? 2 + j : 0)];
                if (tb.p != b)
                    throw std::logic_error("internal: Bsk prime order");
                rd.floor_G1m_top[0][j] = mont(mulmod(rd.floor_G1[j], tb.inv_n, b), b);
                rd.floor_G1m_top[1][j] = mont(mulmod(rd.floor_G1[j], tb.inv_n_w, b), b);
                rd.floor_G1m_top[2][j] = mont(rd.floor_G1m_top[0][j], b);
                rd.floor_G1m_top[3][j] = mont(rd.floor_G1m_top[1][j], b);
                rd.b_p[j] = b;
                rd.b_w1[j] = tb.fwd.size() >= 4 ? tb.fwd[2] : 0; // pair 1 of the forward table: the top layer's twiddle
                rd.b_w1s[j] = tb.fwd.size() >= 4 ? tb.fwd[3] : 0;
                rd.b_rdp[j] = tb.rdp ? tb.rdp : shoup(1, b);
                {
                    u64 inv = b;
                    for (int it = 0; it < 6; it++)
                        inv *= 2 - b * inv;
                    rd.b_ninv[j] = 0 - inv;
                }
            }
            for (int i = 0; i < k; i++)
            {
                const u64 qi = hr.q[i];
                rd.floor_F0_top[0][i] = mulmod(rd.floor_F0[i], tables[i].inv_n, qi);
                rd.floor_F0_top[1][i] = mulmod(rd.floor_F0[i], tables[i].inv_n_w, qi);
                rd.floor_F0_top_s[0][i] = shoup(rd.floor_F0_top[0][i], qi);
                rd.floor_F0_top_s[1][i] = shoup(rd.floor_F0_top[1][i], qi);
                for (int hh = 0; hh < 2; hh++)
                {
                    rd.floor_F0_top[2 + hh][i] = mont(rd.floor_F0_top[hh][i], qi);
                    rd.floor_F0_top_s[2 + hh][i] = shoup(rd.floor_F0_top[2 + hh][i], qi);
                }
                rd.q_p[i] = qi;
                rd.q_rdp[i] = tables[i].rdp ? tables[i].rdp : shoup(1, qi);
                {
                    u64 inv = qi;
                    for (int it = 0; it < 6; it++)
                        inv *= 2 - qi * inv;
                    rd.q_ninv[i] = 0 - inv;
                }
                rd.q_mt_inv_s[i] = shoup(rd.q_mt_inv[i], qi);
                rd.floor_F0_s[i] = shoup(rd.floor_F0[i], qi);
                rd.pBm[i] = mont(hr.prod_B_mod_q[i], qi);
                rd.nBm[i] = mont(qi - hr.prod_B_mod_q[i], qi);
                for (int j = 0; j < B; j++)
                    BQm[static_cast<std::size_t>(i) * B + j] = mont(hr.B_to_q.matrix[static_cast<std::size_t>(i) * B + j], qi);
            }
            for (int j = 0; j < B; j++)
                rd.B_to_mskm[j] = mont(hr.B_to_m_sk.matrix[j], hr.m_sk);
            rd.inv_prod_B_mod_msk_s = shoup(hr.inv_prod_B_mod_m_sk, hr.m_sk);
            rd.lift_L1m = upload<u64>(*this, lt.owned, L1m.data(), L1m.size());
            rd.floor_G2m = upload<u64>(*this, lt.owned, G2m.data(), G2m.size());
            rd.B_to_qm = upload<u64>(*this, lt.owned, BQm.data(), BQm.size());
            // decrypt_scale_and_round constants (rns.cpp:690-716): base q -> {t, gamma}
            {
                HostBaseConv tg;
                tg.build(hr.q, std::vector<u64>{ t, hr.gamma });
                for (int i = 0; i < k; i++)
                {
                    const u64 qi = hr.q[i];
                    rd.dsr_scale[i] = mulmod(mulmod(t % qi, hr.gamma % qi, qi), tg.inv_punct[i], qi);
                    rd.dsr_scale_s[i] = shoup(rd.dsr_scale[i], qi);
                    rd.dsr_to_t[i] = tg.matrix[i];
                    rd.dsr_to_g[i] = tg.matrix[static_cast<std::size_t>(k) + i];
                }
                u64 pq_t = 1 % t, pq_g = 1;
                for (int i = 0; i < k; i++)
                {
                    pq_t = mulmod(pq_t, hr.q[i] % t, t);
                    pq_g = mulmod(pq_g, hr.q[i] % hr.gamma, hr.gamma);
                }
                u64 it = 0, ig = 0, igt = 0;
                if (!invmod(pq_t, t, it) || !invmod(pq_g, hr.gamma, ig) || !invmod(hr.gamma % t, t, igt))
                    throw std::logic_error("invalid rns bases"); // rns.cpp:693-713
                rd.dsr_neg_inv_q_t = it ? t - it : 0;
                rd.dsr_neg_inv_q_g = ig ? hr.gamma - ig : 0;
                rd.dsr_inv_gamma_t = igt;
                rd.dsr_gamma = hr.gamma;
                rd.gamma_prime = static_cast<unsigned>(n_key + 1); // aux primes: m_sk, gamma, B...
            }
            // REDC lands below 2p iff (sum of the bounds of the variable factors) <= 2^64:
            //   lift rows:   k terms t_i < q_i plus temp < b_j;  floor Bsk rows: in < b_j plus k terms < q_i;
            //   conv_sk: B terms < b;  out rows: B terms tb_j < b_j plus alpha-term < m_sk
            const u128 lim = static_cast<u128>(1) << 64;
            // (with a deferred top layer the Bsk operand of the floor rows is a lazy value below 2 b_j)
            const u128 s1 = static_cast<u128>(k) * max_q + 2 * static_cast<u128>(max_b);
            const u128 s2 = static_cast<u128>(B + 1) * max_b;
            rd.redc_small = (s1 <= lim && s2 <= lim) ? 1 : 0;
        }
        lt.h_rns = rd;
        lt.d_rns = upload<RnsDev>(*this, lt.owned, &rd, 1);

        // ---- key-switch constants (only for ciphertext levels) ----
        if (k <= k_first)
        {
            KsDev kd{};
            kd.k = k;
            kd.nsp = nsp;
            kd.nd = (k + nsp - 1) / nsp;
            kd.n_all = k_first;
            kd.n_total = n_key;
            kd.is_ckks = scheme == 2;
            kd.strict = mode_strict;
            const int rows = k + nsp;
            for (int r = 0; r < rows; r++)
                kd.row_prime[r] = static_cast<unsigned>(r < k ? r : k_first + (r - k));
            auto prime_of = [&](int r) { return key_moduli[kd.row_prime[r]]; };
            // mod-up (multi_special_primes.cpp:110-126)
            const std::size_t blk = static_cast<std::size_t>(2 * nsp + rows * nsp);
            std::vector<u64> modup(blk * kd.nd, 0);
            for (int j = 0; j < kd.nd; j++)
            {
                const int r0 = j * nsp, r1 = std::min(r0 + nsp, k), bs = r1 - r0;
                u64 *b = modup.data() + blk * j;
                for (int a = 0; a < bs; a++)
                {
                    const u64 pa = prime_of(r0 + a);
                    u64 inv_prod = 1 % pa;
                    for (int c = 0; c < bs; c++)
                        if (c != a)
                            inv_prod = mulmod(inv_prod, prime_of(r0 + c) % pa, pa);
                    u64 inv = 1;
                    if (bs > 1 && !invmod(inv_prod, pa, inv))
                        throw std::logic_error("modup: inverse does not exist");
                    b[a] = inv;
                    b[nsp + a] = shoup(inv, pa);
                    for (int r = 0; r < rows; r++)
                    {
                        if (r >= r0 && r < r1)
                            continue;
                        const u64 pd = prime_of(r);
                        u64 prod = 1 % pd;
                        for (int c = 0; c < bs; c++)
                            if (c != a)
                                prod = mulmod(prod, prime_of(r0 + c) % pd, pd);
                        b[2 * nsp + r * nsp + a] = prod;
                    }
                }
            }
            kd.modup = upload<u64>(*this, lt.owned, modup.data(), modup.size());
            // mod-down (multi_special_primes.cpp:186-234, :292-299)
            std::vector<u64> neg_hat(static_cast<std::size_t>(k) * nsp, 0);
            for (int j = 0; j < nsp; j++)
            {
                const u64 pj = key_moduli[k_first + j];
                u64 prod = 1 % pj;
                for (int l = 0; l < nsp; l++)
                    if (l != j)
                        prod = mulmod(prod, key_moduli[k_first + l] % pj, pj);
                u64 inv = 1;
                if (nsp > 1 && !invmod(prod, pj, inv))
                    throw std::logic_error("moddown: inverse does not exist");
                kd.inv_hat[j] = inv;
                kd.inv_hat_shoup[j] = shoup(inv, pj);
            }
            for (int i = 0; i < k; i++)
            {
                const u64 qi = key_moduli[i];
                u64 P = 1 % qi;
                for (int j = 0; j < nsp; j++)
                {
                    P = mulmod(P, key_moduli[k_first + j] % qi, qi);
                    u64 prod = 1 % qi;
                    for (int l = 0; l < nsp; l++)
                        if (l != j)
                            prod = mulmod(prod, key_moduli[k_first + l] % qi, qi);
                    neg_hat[static_cast<std::size_t>(i) * nsp + j] = prod ? qi - prod : 0;
                }
                u64 invP;
                if (!invmod(P, qi, invP))
                    throw std::logic_error("moddown: inverse does not exist");
                kd.invP[i] = invP;
                kd.invP_shoup[i] = shoup(invP, qi);
            }
            kd.neg_hat = upload<u64>(*this, lt.owned, neg_hat.data(), neg_hat.size());
            lt.h_ks = kd;
            lt.d_ks = upload<KsDev>(*this, lt.owned, &kd, 1);
        }
        return lt;
    }

    const std::uint32_t *Engine::galois_table(std::uint32_t elt)
    {
        std::lock_guard<std::mutex> lock(mu);
        auto it = galois_tables.find(elt);
        if (it != galois_tables.end())
            return it->second;
        if (lane().capturing)
            throw std::logic_error("the Galois table of this element is not built yet: run the sequence once before capturing");
        // generate_table_ntt, galois.cpp:18-47
        std::vector<std::uint32_t> tab(n);
        const std::uint32_t nm1 = static_cast<std::uint32_t>(n) - 1;
        for (std::size_t i = 0; i < n; i++)
        {
            const std::uint32_t reversed = reverse_bits(static_cast<std::uint32_t>(n + i), logn + 1);
            std::uint64_t index_raw = (static_cast<std::uint64_t>(elt) * reversed) >> 1;
            index_raw &= nm1;
            tab[i] = reverse_bits(static_cast<std::uint32_t>(index_raw), logn);
        }
        SEALHIP_CHECK(hipSetDevice(device));
        std::uint32_t *dev = nullptr;
        SEALHIP_CHECK(hipMalloc(reinterpret_cast<void **>(&dev), sizeof(std::uint32_t) * n));
        SEALHIP_CHECK(hipMemcpy(dev, tab.data(), sizeof(std::uint32_t) * n, hipMemcpyHostToDevice));
        galois_tables.emplace(elt, dev);
        return dev;
    }

    const std::uint32_t *Engine::batch_map()
    {
        std::lock_guard<std::mutex> lock(mu);
        if (d_batch_map)
            return d_batch_map;
        // populate_matrix_reps_index_map, batchencoder.cpp:70-94
        std::vector<std::uint32_t> tab(n);
        const std::size_t row = n >> 1, m = n << 1;
        std::uint64_t pos = 1;
        for (std::size_t i = 0; i < row; i++)
        {
            tab[i] = reverse_bits(static_cast<std::uint32_t>((pos - 1) >> 1), logn);
            tab[row | i] = reverse_bits(static_cast<std::uint32_t>((m - pos - 1) >> 1), logn);
            pos = (pos * 3) & (m - 1);
        }
        // second half of the table: the inverse permutation (encode gathers through it: coalesced stores)
        tab.resize(2 * n);
        for (std::size_t i = 0; i < n; i++)
            tab[n + tab[i]] = static_cast<std::uint32_t>(i);
        SEALHIP_CHECK(hipSetDevice(device));
        SEALHIP_CHECK(hipMalloc(reinterpret_cast<void **>(&d_batch_map), sizeof(std::uint32_t) * 2 * n));
        SEALHIP_CHECK(hipMemcpy(d_batch_map, tab.data(), sizeof(std::uint32_t) * 2 * n, hipMemcpyHostToDevice));
        return d_batch_map;
    }

    void Engine::ckks_tables()
    {
        std::lock_guard<std::mutex> lock(mu);
        if (d_ckks_map)
            return;
        const std::size_t slots = n >> 1, m = n << 1;
        std::vector<std::uint32_t> tab(2 * n);
        std::uint64_t pos = 1;
        for (std::size_t i = 0; i < slots; i++) // ckks.cpp:39-56, generator 5
        {
            tab[i] = reverse_bits(static_cast<std::uint32_t>((pos - 1) >> 1), logn);
            tab[slots | i] = reverse_bits(static_cast<std::uint32_t>((m - pos - 1) >> 1), logn);
            pos = (pos * 5) & (m - 1);
        }
        for (std::size_t i = 0; i < n; i++)
            tab[n + tab[i]] = static_cast<std::uint32_t>(i);
        std::vector<double> roots(2 * n), inv(2 * n);
        for (std::size_t i = 0; i < n; i++) // :62-69
        {
            double re, im;
            complex_root(m, reverse_bits(static_cast<std::uint32_t>(i), logn), re, im);
            roots[2 * i] = re;
            roots[2 * i + 1] = im;
            inv[2 * i] = re;
            inv[2 * i + 1] = -im;
        }
        SEALHIP_CHECK(hipSetDevice(device));
        SEALHIP_CHECK(hipMalloc(reinterpret_cast<void **>(&d_ckks_roots), sizeof(double) * 2 * n));
        SEALHIP_CHECK(hipMalloc(reinterpret_cast<void **>(&d_ckks_inv_roots), sizeof(double) * 2 * n));
        SEALHIP_CHECK(hipMemcpy(d_ckks_roots, roots.data(), sizeof(double) * 2 * n, hipMemcpyHostToDevice));
        SEALHIP_CHECK(hipMemcpy(d_ckks_inv_roots, inv.data(), sizeof(double) * 2 * n, hipMemcpyHostToDevice));
        std::uint32_t *dm = nullptr;
        SEALHIP_CHECK(hipMalloc(reinterpret_cast<void **>(&dm), sizeof(std::uint32_t) * 2 * n));
        SEALHIP_CHECK(hipMemcpy(dm, tab.data(), sizeof(std::uint32_t) * 2 * n, hipMemcpyHostToDevice));
        d_ckks_map = dm;
    }

    namespace
    {
        std::vector<u64> big_product(const std::vector<u64> &moduli, int k)
        {
            std::vector<u64> q(static_cast<std::size_t>(k), 0);
            q[0] = 1;
            for (int i = 0; i < k; i++)
            {
                u128 carry = 0;
                for (int l = 0; l < k; l++)
                {
                    const u128 v = static_cast<u128>(q[l]) * moduli[i] + carry;
                    q[l] = static_cast<u64>(v);
                    carry = v >> 64;
                }
            }
            return q;
        }
    } // namespace

    int Engine::total_coeff_modulus_bit_count(int k)
    {
        std::lock_guard<std::mutex> lock(mu);
        auto it = total_bits.find(k);
        if (it != total_bits.end())
            return it->second;
        const std::vector<u64> q = big_product(key_moduli, k);
        int bits = 0;
        for (int l = k; l-- > 0;)
            if (q[l])
            {
                bits = 64 * l + 64 - __builtin_clzll(q[l]);
                break;
            }
        total_bits[k] = bits;
        return bits;
    }

    const CkksDecodeDev *Engine::ckks_decode_consts(int k)
    {
        if (k < 1 || k > kCkksMaxLimbs || k > n_key)
            throw std::invalid_argument("CKKS decode supports at most 32 coefficient-modulus primes");
        std::lock_guard<std::mutex> lock(mu);
        auto it = ckks_decode.find(k);
        if (it != ckks_decode.end())
            return it->second;
        auto h = std::make_unique<CkksDecodeDev>();
        std::memset(h.get(), 0, sizeof(CkksDecodeDev));
        h->k = k;
        const std::vector<u64> q = big_product(key_moduli, k);
        u128 carry = 1; // (q + 1) >> 1
        std::vector<u64> plus(static_cast<std::size_t>(k) + 1, 0);
        for (int l = 0; l < k; l++)
        {
            carry += q[l];
            plus[l] = static_cast<u64>(carry);
            carry >>= 64;
        }
        plus[k] = static_cast<u64>(carry);
        for (int l = 0; l < k; l++)
        {
            h->q[l] = q[l];
            h->half[l] = (plus[l] >> 1) | (plus[l + 1] << 63);
        }
        for (int i = 0; i < k; i++)
        {
            const u64 qi = key_moduli[i];
            u128 rem = 0;
            for (int l = k; l-- > 0;) // q / q_i
            {
                const u128 cur = (rem << 64) | q[l];
                h->punct[i * kCkksMaxLimbs + l] = static_cast<u64>(cur / qi);
                rem = cur % qi;
            }
            u128 r = 0;
            for (int l = k; l-- > 0;)
                r = ((r << 64) | h->punct[i * kCkksMaxLimbs + l]) % qi;
            if (!invmod(static_cast<u64>(r), qi, h->inv_punct[i]))
                throw std::invalid_argument("coefficient moduli are not coprime");
        }
        SEALHIP_CHECK(hipSetDevice(device));
        CkksDecodeDev *dev = nullptr;
        SEALHIP_CHECK(hipMalloc(reinterpret_cast<void **>(&dev), sizeof(CkksDecodeDev)));
        SEALHIP_CHECK(hipMemcpy(dev, h.get(), sizeof(CkksDecodeDev), hipMemcpyHostToDevice));
        ckks_decode.emplace(k, dev);
        return dev;
    }

    void Engine::ws_reserve(std::size_t bytes) const
    {
        Lane &l = lane();
        if (bytes <= l.ws_bytes)
            return;
        if (l.capturing)
            throw std::logic_error("the workspace would grow during a graph capture: run the sequence once before capturing");
        SEALHIP_CHECK(hipSetDevice(device));
        if (l.ws)
        {
            SEALHIP_CHECK(hipStreamSynchronize(l.stream));
            SEALHIP_CHECK(hipFree(l.ws));
            l.ws = nullptr;
            l.ws_bytes = 0;
        }
        SEALHIP_CHECK(hipMalloc(&l.ws, bytes));
        l.ws_bytes = bytes;
        l.alloc_generation++;
    }

    u64 *Engine::ws_alloc(std::size_t words) const
    {
        Lane &l = lane();
        const std::size_t bytes = (words * sizeof(u64) + 255) & ~static_cast<std::size_t>(255);
        if (l.ws_used + bytes > l.ws_bytes)
            throw std::logic_error("internal: workspace overflow");
        u64 *p = reinterpret_cast<u64 *>(static_cast<char *>(l.ws) + l.ws_used);
        l.ws_used += bytes;
        return p;
    }

    int Engine::rows_for(int k, unsigned base)
    {
        LevelTools &lt = level_host(k);
        switch (base)
        {
        case 0:
            return lt.map_q.rows;
        case 1:
            return lt.map_bsk.rows;
        case 2:
            return lt.map_key.rows;
        default:
            throw std::invalid_argument("unknown base");
        }
    }

    RowMap Engine::map_for(int k, unsigned base)
    {
        LevelTools &lt = level_host(k);
        switch (base)
        {
        case 0:
            return lt.map_q;
        case 1:
            if (scheme != 1)
                throw std::invalid_argument("base Bsk exists only for BFV");
            return lt.map_bsk;
        case 2:
            if (k > k_first)
                throw std::invalid_argument("key base needs a ciphertext level");
            return lt.map_key;
        default:
            throw std::invalid_argument("unknown base");
        }
    }
} // namespace sealhip
