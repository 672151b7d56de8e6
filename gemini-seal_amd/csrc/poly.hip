// poly.hip -- coefficient-wise kernels (native/src/seal/util/polyarithsmallmod.{h,cpp}) and the
// Galois automorphisms (native/src/seal/util/galois.cpp) over batches of RNS rows.
// All of these are pure HBM streaming: 16-byte loads/stores, two coefficients per lane.
#include "engine.hpp"

namespace sealhip
{
    namespace
    {
        constexpr int kThreads = 256;

        inline unsigned grid_for(std::size_t work_items)
        {
            std::size_t blocks = (work_items + kThreads - 1) / kThreads;
            const std::size_t cap = 256u * 16u; // grid-stride the rest
            return static_cast<unsigned>(blocks < cap ? (blocks ? blocks : 1) : cap);
        }

        template <int OP>
        __device__ __forceinline__ u64 apply_op(u64 a, u64 b, u64 scalar, const PrimeDev &P)
        {
            if (OP == 0)
                return mul_mod(a, b, P.p, P.cr0, P.cr1); // dyadic_product_coeffmod, polyarithsmallmod.cpp:63-117
            if (OP == 1)
                return add_mod(a, b, P.p);
            if (OP == 2)
                return sub_mod(a, b, P.p);
            if (OP == 3)
                return neg_mod(a, P.p);
            if (OP == 5)
                return barrett_reduce_63(a, P.p, P.cr1); // modulo_poly_coeffs_63, polyarithsmallmod.h:98-120
            return mul_mod(a, scalar, P.p, P.cr0, P.cr1); // multiply_poly_scalar_coeffmod, :15-61
        }

        template <int OP>
        __global__ __launch_bounds__(kThreads) void poly_op_kernel(const u64 *__restrict__ a,
                                                                   const u64 *__restrict__ b, u64 scalar,
                                                                   u64 *__restrict__ r,
                                                                   const PrimeDev *__restrict__ primes, RowMap map,
                                                                   int logn, std::size_t npairs)
        {
            const std::size_t stride = static_cast<std::size_t>(gridDim.x) * blockDim.x;
            for (std::size_t i = blockIdx.x * static_cast<std::size_t>(blockDim.x) + threadIdx.x; i < npairs;
                 i += stride)
            {
                const std::size_t row = (2 * i) >> logn;
                const unsigned short pid = map.prime[row % map.rows];
                if (pid == kSkipRow)
                    continue;
                const PrimeDev &P = primes[pid];
                const ulonglong2 va = reinterpret_cast<const ulonglong2 *>(a)[i];
                ulonglong2 vb = va;
                if (OP == 0 || OP == 1 || OP == 2)
                    vb = reinterpret_cast<const ulonglong2 *>(b)[i];
                ulonglong2 out;
                out.x = apply_op<OP>(va.x, vb.x, scalar, P);
                out.y = apply_op<OP>(va.y, vb.y, scalar, P);
                reinterpret_cast<ulonglong2 *>(r)[i] = out;
            }
        }

        // out[I] = sum over i1 + i2 = I of a[i1] (.) b[i2], each term reduced (dyadic_product_coeffmod) and
        // accumulated with add_poly_coeffmod, exactly as evaluator.cpp:376-420 / :493-520 do.
        __global__ __launch_bounds__(kThreads) void tensor_product_kernel(
            const u64 *__restrict__ a, int sa, std::size_t a_stride, const u64 *__restrict__ b, int sb,
            std::size_t b_stride, u64 *__restrict__ out, std::size_t out_stride, const PrimeDev *__restrict__ primes,
            RowMap map, int logn, std::size_t npairs_per_item, std::size_t count, unsigned *__restrict__ tflags, int square)
        {
            const std::size_t total = npairs_per_item * count;
            const std::size_t stride = static_cast<std::size_t>(gridDim.x) * blockDim.x;
            const std::size_t poly_words = static_cast<std::size_t>(map.rows) << logn;
            for (std::size_t i = blockIdx.x * static_cast<std::size_t>(blockDim.x) + threadIdx.x; i < total;
                 i += stride)
            {
                const std::size_t item = i / npairs_per_item;
                const std::size_t off = 2 * (i - item * npairs_per_item); // word offset inside one polynomial
                const PrimeDev &P = primes[map.prime[off >> logn]];
                const u64 *pa = a + item * a_stride + off;
                const u64 *pb = b + item * b_stride + off;
                u64 *po = out + item * out_stride + off;
                if (square)
                {
                    // Evaluator::square on a size-2 operand (bfv_square evaluator.cpp:644-657, ckks_square :752-760): x_0^2, x_0 x_1
                    // added to itself, x_1^2 -- two polynomials read instead of four, three products instead of four
                    const ulonglong2 a0 = *reinterpret_cast<const ulonglong2 *>(pa);
                    const ulonglong2 a1 = *reinterpret_cast<const ulonglong2 *>(pa + poly_words);
                    ulonglong2 c0, c1, c2;
                    c0.x = mul_mod(a0.x, a0.x, P.p, P.cr0, P.cr1);
                    c0.y = mul_mod(a0.y, a0.y, P.p, P.cr0, P.cr1);
                    c1.x = mul_mod(a0.x, a1.x, P.p, P.cr0, P.cr1);
                    c1.y = mul_mod(a0.y, a1.y, P.p, P.cr0, P.cr1);
                    c1.x = add_mod(c1.x, c1.x, P.p);
                    c1.y = add_mod(c1.y, c1.y, P.p);
                    c2.x = mul_mod(a1.x, a1.x, P.p, P.cr0, P.cr1);
                    c2.y = mul_mod(a1.y, a1.y, P.p, P.cr0, P.cr1);
                    *reinterpret_cast<ulonglong2 *>(po) = c0;
                    *reinterpret_cast<ulonglong2 *>(po + poly_words) = c1;
                    *reinterpret_cast<ulonglong2 *>(po + 2 * poly_words) = c2;
                    note_nonzero(tflags, item, c1.x | c1.y | c2.x | c2.y);
                    continue;
                }
                if (sa == 2 && sb == 2)
                {
                    const ulonglong2 a0 = *reinterpret_cast<const ulonglong2 *>(pa);
                    const ulonglong2 a1 = *reinterpret_cast<const ulonglong2 *>(pa + poly_words);
                    const ulonglong2 b0 = *reinterpret_cast<const ulonglong2 *>(pb);
                    const ulonglong2 b1 = *reinterpret_cast<const ulonglong2 *>(pb + poly_words);
                    ulonglong2 c0, c1, c2;
                    c0.x = mul_mod(a0.x, b0.x, P.p, P.cr0, P.cr1);
                    c0.y = mul_mod(a0.y, b0.y, P.p, P.cr0, P.cr1);
                    c1.x = add_mod(mul_mod(a1.x, b0.x, P.p, P.cr0, P.cr1), mul_mod(a0.x, b1.x, P.p, P.cr0, P.cr1), P.p);
                    c1.y = add_mod(mul_mod(a1.y, b0.y, P.p, P.cr0, P.cr1), mul_mod(a0.y, b1.y, P.p, P.cr0, P.cr1), P.p);
                    c2.x = mul_mod(a1.x, b1.x, P.p, P.cr0, P.cr1);
                    c2.y = mul_mod(a1.y, b1.y, P.p, P.cr0, P.cr1);
                    *reinterpret_cast<ulonglong2 *>(po) = c0;
                    *reinterpret_cast<ulonglong2 *>(po + poly_words) = c1;
                    *reinterpret_cast<ulonglong2 *>(po + 2 * poly_words) = c2;
                    note_nonzero(tflags, item, c1.x | c1.y | c2.x | c2.y);
                    continue;
                }
                const int dest = sa + sb - 1;
                for (int I = 0; I < dest; I++)
                {
                    const int last1 = I < sa - 1 ? I : sa - 1;
                    const int first2 = I < sb - 1 ? I : sb - 1;
                    const int first1 = I - first2;
                    ulonglong2 acc;
                    acc.x = 0;
                    acc.y = 0;
                    for (int i1 = first1; i1 <= last1; i1++)
                    {
                        const int i2 = I - i1;
                        const ulonglong2 va = *reinterpret_cast<const ulonglong2 *>(pa + i1 * poly_words);
                        const ulonglong2 vb = *reinterpret_cast<const ulonglong2 *>(pb + i2 * poly_words);
                        acc.x = add_mod(mul_mod(va.x, vb.x, P.p, P.cr0, P.cr1), acc.x, P.p);
                        acc.y = add_mod(mul_mod(va.y, vb.y, P.p, P.cr0, P.cr1), acc.y, P.p);
                    }
                    *reinterpret_cast<ulonglong2 *>(po + I * poly_words) = acc;
                    if (I >= 1)
                        note_nonzero(tflags, item, acc.x | acc.y);
                }
            }
        }

        __global__ __launch_bounds__(kThreads) void copy_rows_kernel(const u64 *__restrict__ src,
                                                                     std::size_t src_stride, u64 *__restrict__ dst,
                                                                     std::size_t dst_stride,
                                                                     std::size_t pairs_per_poly, std::size_t npolys)
        {
            const std::size_t total = pairs_per_poly * npolys;
            const std::size_t stride = static_cast<std::size_t>(gridDim.x) * blockDim.x;
            for (std::size_t i = blockIdx.x * static_cast<std::size_t>(blockDim.x) + threadIdx.x; i < total;
                 i += stride)
            {
                const std::size_t poly = i / pairs_per_poly;
                const std::size_t off = 2 * (i - poly * pairs_per_poly);
                *reinterpret_cast<ulonglong2 *>(dst + poly * dst_stride + off) =
                    *reinterpret_cast<const ulonglong2 *>(src + poly * src_stride + off);
            }
        }

        // NTT form: out[i] = in[table[i]] (galois.cpp:188-214); coefficient form: out[(i*g) mod N] = +-in[i]
        // (galois.cpp:144-186). One coefficient per lane: the permutation defeats wider accesses on one side.
        __global__ __launch_bounds__(kThreads) void galois_kernel(const u64 *__restrict__ in, u64 *__restrict__ out,
                                                                  const PrimeDev *__restrict__ primes, RowMap map,
                                                                  int logn, std::size_t total, std::uint32_t elt,
                                                                  const std::uint32_t *__restrict__ table)
        {
            const std::size_t stride = static_cast<std::size_t>(gridDim.x) * blockDim.x;
            const std::size_t nmask = (static_cast<std::size_t>(1) << logn) - 1;
            for (std::size_t i = blockIdx.x * static_cast<std::size_t>(blockDim.x) + threadIdx.x; i < total;
                 i += stride)
            {
                const std::size_t row = i >> logn;
                const std::size_t c = i & nmask;
                if (table)
                {
                    out[i] = in[(row << logn) + table[c]];
                }
                else
                {
                    const u64 p = primes[map.prime[row % map.rows]].p;
                    const u64 raw = static_cast<u64>(c) * elt;
                    u64 v = in[i];
                    if ((raw >> logn) & 1)
                        v = neg_mod(v, p);
                    out[(row << logn) + (raw & nmask)] = v;
                }
            }
        }

        // Evaluator-level linear operations on batches of ciphertexts whose sizes may differ
        // (evaluator.cpp:65-232: negate_inplace, add_inplace, sub_inplace) and multiply_plain_ntt (:1605-1646).
        // OP 0: a + b (missing polynomials of the shorter operand count as absent: the tail of the longer one is
        // copied), OP 1: a - b (tail of b negated, :216-220), OP 2: -a, OP 3: a (.) plain, the plaintext rows
        // broadcast over the polynomials of the ciphertext.
        template <int OP>
        __global__ __launch_bounds__(kThreads) void ct_linear_kernel(const u64 *__restrict__ a, int sa,
                                                                     const u64 *__restrict__ b, int sb,
                                                                     std::size_t b_item_stride, u64 *__restrict__ out,
                                                                     const PrimeDev *__restrict__ primes, RowMap map,
                                                                     int logn, std::size_t pairs_per_poly,
                                                                     std::size_t count)
        {
            const int so = OP <= 1 ? (sa > sb ? sa : sb) : sa;
            const std::size_t total = pairs_per_poly * so * count;
            const std::size_t stride = static_cast<std::size_t>(gridDim.x) * blockDim.x;
            const std::size_t poly_words = pairs_per_poly * 2;
            for (std::size_t i = blockIdx.x * static_cast<std::size_t>(blockDim.x) + threadIdx.x; i < total;
                 i += stride)
            {
                const std::size_t item = i / (pairs_per_poly * so);
                const std::size_t rem = i - item * pairs_per_poly * so;
                const int j = static_cast<int>(rem / pairs_per_poly);
                const std::size_t off = 2 * (rem - j * pairs_per_poly);
                const PrimeDev &P = primes[map.prime[off >> logn]];
                ulonglong2 va, vb, r;
                va.x = va.y = vb.x = vb.y = 0;
                const bool has_a = j < sa;
                if (has_a)
                    va = *reinterpret_cast<const ulonglong2 *>(a + (item * sa + j) * poly_words + off);
                if (OP <= 1)
                {
                    const bool has_b = j < sb;
                    if (has_b)
                        vb = *reinterpret_cast<const ulonglong2 *>(b + (item * sb + j) * poly_words + off);
                    if (has_a && has_b)
                    {
                        r.x = OP == 0 ? add_mod(va.x, vb.x, P.p) : sub_mod(va.x, vb.x, P.p);
                        r.y = OP == 0 ? add_mod(va.y, vb.y, P.p) : sub_mod(va.y, vb.y, P.p);
                    }
                    else if (has_a)
                        r = va;
                    else
                    {
                        r.x = OP == 0 ? vb.x : neg_mod(vb.x, P.p);
                        r.y = OP == 0 ? vb.y : neg_mod(vb.y, P.p);
                    }
                }
                else if (OP == 2)
                {
                    r.x = neg_mod(va.x, P.p);
                    r.y = neg_mod(va.y, P.p);
                }
                else
                {
                    vb = *reinterpret_cast<const ulonglong2 *>(b + item * b_item_stride + off);
                    r.x = mul_mod(va.x, vb.x, P.p, P.cr0, P.cr1);
                    r.y = mul_mod(va.y, vb.y, P.p, P.cr0, P.cr1);
                }
                *reinterpret_cast<ulonglong2 *>(out + (item * so + j) * poly_words + off) = r;
            }
        }

        // Ciphertext::is_transparent (ciphertext.h:471-476): flag[item] != 0 iff some word of polynomials 1.. is non-zero
        __global__ __launch_bounds__(kThreads) void nonzero_tail_kernel(const u64 *__restrict__ ct,
                                                                        std::size_t item_words,
                                                                        std::size_t skip_words, std::size_t count,
                                                                        unsigned *__restrict__ flag)
        {
            const std::size_t tail_pairs = (item_words - skip_words) / 2;
            const std::size_t total = tail_pairs * count;
            const std::size_t stride = static_cast<std::size_t>(gridDim.x) * blockDim.x;
            for (std::size_t i = blockIdx.x * static_cast<std::size_t>(blockDim.x) + threadIdx.x; i < total;
                 i += stride)
            {
                const std::size_t item = i / tail_pairs;
                const std::size_t off = 2 * (i - item * tail_pairs);
                const ulonglong2 v = *reinterpret_cast<const ulonglong2 *>(ct + item * item_words + skip_words + off);
                if ((v.x | v.y) != 0 && flag[item] == 0)
                    flag[item] = 1; // benign race: every writer stores the same value
            }
        }

        // is_data_valid_for (valcheck.cpp:284-317): flag[item] != 0 iff some coefficient is not below its row's prime
        __global__ __launch_bounds__(kThreads) void out_of_range_kernel(const u64 *__restrict__ ct, std::size_t item_words,
                                                                        std::size_t count,
                                                                        const PrimeDev *__restrict__ primes, RowMap map,
                                                                        int logn, unsigned *__restrict__ flag)
        {
            const std::size_t item_pairs = item_words / 2;
            const std::size_t total = item_pairs * count;
            const std::size_t stride = static_cast<std::size_t>(gridDim.x) * blockDim.x;
            for (std::size_t i = blockIdx.x * static_cast<std::size_t>(blockDim.x) + threadIdx.x; i < total;
                 i += stride)
            {
                const std::size_t item = i / item_pairs;
                const std::size_t off = 2 * (i - item * item_pairs);
                const u64 p = primes[map.prime[(off >> logn) % map.rows]].p;
                const ulonglong2 v = *reinterpret_cast<const ulonglong2 *>(ct + item * item_words + off);
                if ((v.x >= p || v.y >= p) && flag[item] == 0)
                    flag[item] = 1; // benign race: every writer stores the same value
            }
        }

        // multiply_plain_normal's lift of a plaintext into the RNS base when every q_i > t (fast plain lift,
        // evaluator.cpp:1583-1592): temp_r[i] = plain[i] + (plain[i] >= threshold ? q_r - t : 0)
        __global__ __launch_bounds__(kThreads) void plain_lift_kernel(const u64 *__restrict__ plain,
                                                                      std::size_t plain_stride, u64 *__restrict__ out,
                                                                      const PrimeDev *__restrict__ primes, RowMap map,
                                                                      int logn, u64 t, u64 threshold,
                                                                      std::size_t nplains)
        {
            const std::size_t n = static_cast<std::size_t>(1) << logn;
            const std::size_t total = nplains * map.rows * n;
            const std::size_t stride = static_cast<std::size_t>(gridDim.x) * blockDim.x;
            for (std::size_t i = blockIdx.x * static_cast<std::size_t>(blockDim.x) + threadIdx.x; i < total;
                 i += stride)
            {
                const std::size_t c = i & (n - 1);
                const std::size_t row = i >> logn;
                const std::size_t item = row / map.rows;
                const u64 p = primes[map.prime[row % map.rows]].p;
                const u64 v = plain[item * plain_stride + c];
                out[i] = v + (v >= threshold ? p - t : 0);
            }
        }

        // Decryptor::dot_product_ct_sk_array (decryptor.cpp:246-256, :265): out = sum_{i>=1} ct_i (.) s^i, each term
        // reduced (dyadic_product_coeffmod) and accumulated with add_poly_coeffmod; with add_c0 also + ct_0 (NTT-form
        // ciphertexts; coefficient-form ones add c_0 after the inverse NTT). ct polys 1.. may hold lazy NTT values.
        __global__ __launch_bounds__(kThreads) void dot_sk_kernel(const u64 *__restrict__ ct, int size,
                                                                  std::size_t ct_item_stride,
                                                                  const u64 *__restrict__ sk, std::size_t sk_power_stride,
                                                                  u64 *out,
                                                                  const PrimeDev *__restrict__ primes, RowMap map, int logn,
                                                                  std::size_t pairs_per_poly, std::size_t count, int add_c0)
        {
            const std::size_t total = pairs_per_poly * count;
            const std::size_t stride = static_cast<std::size_t>(gridDim.x) * blockDim.x;
            const std::size_t poly_words = pairs_per_poly * 2;
            for (std::size_t i = blockIdx.x * static_cast<std::size_t>(blockDim.x) + threadIdx.x; i < total; i += stride)
            {
                const std::size_t item = i / pairs_per_poly;
                const std::size_t off = 2 * (i - item * pairs_per_poly);
                const PrimeDev &P = primes[map.prime[off >> logn]];
                ulonglong2 acc;
                acc.x = acc.y = 0;
                if (add_c0 == 2)
                    acc = *reinterpret_cast<const ulonglong2 *>(out + item * poly_words + off);
                for (int j = 1; j < size; j++)
                {
                    const ulonglong2 c = *reinterpret_cast<const ulonglong2 *>(ct + item * ct_item_stride + j * poly_words + off);
                    const ulonglong2 s = *reinterpret_cast<const ulonglong2 *>(sk + (j - 1) * sk_power_stride + off);
                    acc.x = add_mod(acc.x, mul_mod(c.x, s.x, P.p, P.cr0, P.cr1), P.p);
                    acc.y = add_mod(acc.y, mul_mod(c.y, s.y, P.p, P.cr0, P.cr1), P.p);
                }
                if (add_c0)
                {
                    const ulonglong2 c0 = *reinterpret_cast<const ulonglong2 *>(ct + item * ct_item_stride + off);
                    acc.x = add_mod(acc.x, c0.x, P.p);
                    acc.y = add_mod(acc.y, c0.y, P.p);
                }
                *reinterpret_cast<ulonglong2 *>(out + item * poly_words + off) = acc;
            }
        }
    } // namespace

    hipError_t launch_poly_op(const Engine &e, PolyOp op, const u64 *a, const u64 *b, u64 scalar, u64 *r,
                              std::size_t nrows, const RowMap &map)
    {
        const std::size_t npairs = (nrows << e.logn) / 2;
        if (npairs == 0)
            return hipSuccess;
        const unsigned grid = grid_for(npairs);
        ProfScope prof(e, "poly_op", 0);
        switch (op)
        {
        case PolyOp::Dyadic:
            poly_op_kernel<0><<<grid, kThreads, 0, e.lane().stream>>>(a, b, scalar, r, e.d_primes, map, e.logn, npairs);
            break;
        case PolyOp::Add:
            poly_op_kernel<1><<<grid, kThreads, 0, e.lane().stream>>>(a, b, scalar, r, e.d_primes, map, e.logn, npairs);
            break;
        case PolyOp::Sub:
            poly_op_kernel<2><<<grid, kThreads, 0, e.lane().stream>>>(a, b, scalar, r, e.d_primes, map, e.logn, npairs);
            break;
        case PolyOp::Negate:
            poly_op_kernel<3><<<grid, kThreads, 0, e.lane().stream>>>(a, b, scalar, r, e.d_primes, map, e.logn, npairs);
            break;
        case PolyOp::Scalar:
            poly_op_kernel<4><<<grid, kThreads, 0, e.lane().stream>>>(a, b, scalar, r, e.d_primes, map, e.logn, npairs);
            break;
        case PolyOp::Mod63:
            poly_op_kernel<5><<<grid, kThreads, 0, e.lane().stream>>>(a, b, scalar, r, e.d_primes, map, e.logn, npairs);
            break;
        }
        return hipGetLastError();
    }

    hipError_t launch_tensor_product(const Engine &e, const u64 *a, int sa, std::size_t a_stride, const u64 *b, int sb,
                                     std::size_t b_stride, u64 *out, std::size_t out_stride, std::size_t count,
                                     const RowMap &map, bool square)
    {
        if (square && (sa != 2 || sb != 2 || a != b))
            return hipErrorInvalidValue; // the square form is that of ONE size-2 operand
        const std::size_t pairs = (static_cast<std::size_t>(map.rows) << e.logn) / 2;
        if (pairs * count == 0)
            return hipSuccess;
        ProfScope prof(e, "tensor_product", 0);
        tensor_product_kernel<<<grid_for(pairs * count), kThreads, 0, e.lane().stream>>>(
            a, sa, a_stride, b, sb, b_stride, out, out_stride, e.d_primes, map, e.logn, pairs, count, e.lane().tsink_arm,
            square ? 1 : 0);
        return hipGetLastError();
    }

    namespace
    {
        struct RowValues
        {
            u64 v[kMaxModuli];
        };
        // dst[count][rows][N]: row r of every item filled with vals.v[r] (the single-value CKKS encodings, ckks.cpp:80-260)
        __global__ __launch_bounds__(kThreads) void fill_rows_kernel(u64 *__restrict__ dst, RowValues vals, int rows, int logn,
                                                                    std::size_t npairs)
        {
            const std::size_t stride = static_cast<std::size_t>(gridDim.x) * blockDim.x;
            for (std::size_t i = blockIdx.x * static_cast<std::size_t>(blockDim.x) + threadIdx.x; i < npairs; i += stride)
            {
                const u64 v = vals.v[((2 * i) >> logn) % rows];
                ulonglong2 o;
                o.x = v;
                o.y = v;
                reinterpret_cast<ulonglong2 *>(dst)[i] = o;
            }
        }
    } // namespace

    hipError_t launch_fill_rows(const Engine &e, u64 *dst, const u64 *row_values, int rows, std::size_t count)
    {
        if (rows < 1 || rows > kMaxModuli)
            return hipErrorInvalidValue;
        const std::size_t npairs = (count * static_cast<std::size_t>(rows) << e.logn) / 2;
        if (!npairs)
            return hipSuccess;
        RowValues rv{};
        for (int r = 0; r < rows; r++)
            rv.v[r] = row_values[r];
        ProfScope prof(e, "fill_rows", 0);
        fill_rows_kernel<<<grid_for(npairs), kThreads, 0, e.lane().stream>>>(dst, rv, rows, e.logn, npairs);
        return hipGetLastError();
    }

    hipError_t launch_copy_rows(const Engine &e, const u64 *src, std::size_t src_poly_stride, u64 *dst,
                                std::size_t dst_poly_stride, std::size_t npolys, int rows)
    {
        const std::size_t pairs = (static_cast<std::size_t>(rows) << e.logn) / 2;
        if (pairs * npolys == 0)
            return hipSuccess;
        ProfScope prof(e, "copy_rows", 0);
        copy_rows_kernel<<<grid_for(pairs * npolys), kThreads, 0, e.lane().stream>>>(src, src_poly_stride, dst,
                                                                             dst_poly_stride, pairs, npolys);
        return hipGetLastError();
    }

    hipError_t launch_galois(const Engine &e, const u64 *in, u64 *out, std::size_t nrows, const RowMap &map,
                             std::uint32_t elt, const std::uint32_t *table)
    {
        const std::size_t total = nrows << e.logn;
        if (total == 0)
            return hipSuccess;
        ProfScope prof(e, "galois", 0);
        galois_kernel<<<grid_for(total), kThreads, 0, e.lane().stream>>>(in, out, e.d_primes, map, e.logn, total, elt, table);
        return hipGetLastError();
    }
    hipError_t launch_ct_linear(const Engine &e, CtLinearOp op, const u64 *a, int sa, const u64 *b, int sb,
                                std::size_t b_item_stride, u64 *out, std::size_t count, const RowMap &map)
    {
        const std::size_t pairs = (static_cast<std::size_t>(map.rows) << e.logn) / 2;
        const int so = (op == CtLinearOp::Add || op == CtLinearOp::Sub) ? (sa > sb ? sa : sb) : sa;
        if (pairs * so * count == 0)
            return hipSuccess;
        ProfScope prof(e, "ct_linear", 0);
        const unsigned grid = grid_for(pairs * so * count);
        switch (op)
        {
        case CtLinearOp::Add:
            ct_linear_kernel<0><<<grid, kThreads, 0, e.lane().stream>>>(a, sa, b, sb, b_item_stride, out, e.d_primes, map, e.logn,
                                                                pairs, count);
            break;
        case CtLinearOp::Sub:
            ct_linear_kernel<1><<<grid, kThreads, 0, e.lane().stream>>>(a, sa, b, sb, b_item_stride, out, e.d_primes, map, e.logn,
                                                                pairs, count);
            break;
        case CtLinearOp::Negate:
            ct_linear_kernel<2><<<grid, kThreads, 0, e.lane().stream>>>(a, sa, b, sb, b_item_stride, out, e.d_primes, map, e.logn,
                                                                pairs, count);
            break;
        case CtLinearOp::MulPlain:
            ct_linear_kernel<3><<<grid, kThreads, 0, e.lane().stream>>>(a, sa, b, sb, b_item_stride, out, e.d_primes, map, e.logn,
                                                                pairs, count);
            break;
        }
        return hipGetLastError();
    }

    // flags[item] |= "some word of [data + item * item_stride, + words) is non-zero" (the read pass the fused kernels avoid)
    hipError_t launch_nonzero_words(const Engine &e, const u64 *data, std::size_t item_stride, std::size_t words,
                                    std::size_t count, unsigned *flags)
    {
        if (words == 0 || count == 0)
            return hipSuccess;
        if (words > item_stride)
            return hipErrorInvalidValue;
        ProfScope prof(e, "nonzero_tail", 0);
        // (item_words = stride, skip = stride - words, base shifted back so that the tail is [data, data + words))
        nonzero_tail_kernel<<<grid_for(words / 2 * count), kThreads, 0, e.lane().stream>>>(
            data - (item_stride - words), item_stride, item_stride - words, count, flags);
        return hipGetLastError();
    }

    hipError_t launch_nonzero_tail(const Engine &e, const u64 *ct, std::size_t item_words, std::size_t skip_words,
                                   std::size_t count, unsigned *flags)
    {
        if (item_words <= skip_words || count == 0)
            return hipSuccess;
        ProfScope prof(e, "nonzero_tail", 0);
        nonzero_tail_kernel<<<grid_for((item_words - skip_words) / 2 * count), kThreads, 0, e.lane().stream>>>(
            ct, item_words, skip_words, count, flags);
        return hipGetLastError();
    }

    hipError_t launch_out_of_range(const Engine &e, const u64 *ct, std::size_t item_words, std::size_t count,
                                   const RowMap &map, unsigned *flags)
    {
        if (item_words == 0 || count == 0)
            return hipSuccess;
        ProfScope prof(e, "out_of_range", 0);
        out_of_range_kernel<<<grid_for(item_words / 2 * count), kThreads, 0, e.lane().stream>>>(ct, item_words, count, e.d_primes,
                                                                                        map, e.logn, flags);
        return hipGetLastError();
    }

    hipError_t launch_plain_lift(const Engine &e, const u64 *plain, std::size_t plain_stride, u64 *out, std::size_t nplains,
                                 const RowMap &map, u64 t)
    {
        const std::size_t total = (nplains * map.rows) << e.logn;
        if (total == 0)
            return hipSuccess;
        ProfScope prof(e, "plain_lift", 0);
        plain_lift_kernel<<<grid_for(total), kThreads, 0, e.lane().stream>>>(plain, plain_stride, out, e.d_primes, map, e.logn, t,
                                                                    (t + 1) >> 1, nplains);
        return hipGetLastError();
    }
    hipError_t launch_dot_sk(const Engine &e, const u64 *ct, int size, std::size_t ct_item_stride, const u64 *sk_powers,
                             std::size_t sk_power_stride, u64 *out, std::size_t count, const RowMap &map, int add_c0)
    {
        const std::size_t pairs = (static_cast<std::size_t>(map.rows) << e.logn) / 2;
        if (pairs * count == 0)
            return hipSuccess;
        ProfScope prof(e, "dot_sk", 0);
        dot_sk_kernel<<<grid_for(pairs * count), kThreads, 0, e.lane().stream>>>(ct, size, ct_item_stride, sk_powers, sk_power_stride,
                                                                         out, e.d_primes, map, e.logn, pairs, count,
                                                                         add_c0);
        return hipGetLastError();
    }
} // namespace sealhip
