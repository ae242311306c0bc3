// Host-side batching: bit-exact integer work, no GPU involved.
//  * psvae_collate_pairs: the (graph, row, column) walk of process_step3
//    (src/data/bpe_dataset.py:110-132) including its negative-pair sampler, which
//    draws from CPython's Mersenne Twister exactly where the reference calls
//    random(): only for cross-piece, non-bonded ordered pairs.
//  * psvae_pack_batch: padded numbering -> packed rows, CSR by target / by
//    source (stable, so a row keeps the reference's edge order and segment sums
//    add in the same order as index_add_), and the inverse lists that make the
//    scatter-type gradients atomics-free.
#include <stdint.h>
#include <string.h>
#include <vector>
#include "../../include/psvae.h"

namespace {

// MT19937 as CPython's _random module runs it: state = 624 words + position.
inline uint32_t mt_next(uint32_t* st) {
    constexpr int N = 624, M = 397;
    constexpr uint32_t A = 0x9908b0dfu, UP = 0x80000000u, LO = 0x7fffffffu;
    uint32_t pos = st[N];
    if (pos >= (uint32_t)N) {
        int kk = 0;
        uint32_t y;
        for (; kk < N - M; ++kk) { y = (st[kk] & UP) | (st[kk + 1] & LO); st[kk] = st[kk + M] ^ (y >> 1) ^ ((y & 1u) ? A : 0u); }
        for (; kk < N - 1; ++kk) { y = (st[kk] & UP) | (st[kk + 1] & LO); st[kk] = st[kk + (M - N)] ^ (y >> 1) ^ ((y & 1u) ? A : 0u); }
        y = (st[N - 1] & UP) | (st[0] & LO);
        st[N - 1] = st[M - 1] ^ (y >> 1) ^ ((y & 1u) ? A : 0u);
        pos = 0;
    }
    uint32_t y = st[pos++];
    st[N] = pos;
    y ^= y >> 11;
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= y >> 18;
    return y;
}
// random.random(): 53-bit resolution from two draws.
inline double mt_random(uint32_t* st) {
    const uint32_t a = mt_next(st) >> 5, b = mt_next(st) >> 6;
    return (a * 67108864.0 + b) * (1.0 / 9007199254740992.0);
}

// stable counting sort of items by key -> ptr[nkeys+1], order[count]
void bucket(const std::vector<int32_t>& key, int nkeys, int32_t* ptr, int32_t* order) {
    for (int k = 0; k <= nkeys; ++k) ptr[k] = 0;
    for (int32_t v : key) ptr[v + 1]++;
    for (int k = 0; k < nkeys; ++k) ptr[k + 1] += ptr[k];
    std::vector<int32_t> cur(ptr, ptr + nkeys);
    for (size_t i = 0; i < key.size(); ++i) order[cur[key[i]]++] = (int32_t)i;
}

}  // namespace

extern "C" int psvae_collate_pairs(int B, int N, const int32_t* n_atoms, const int8_t* ad, const int8_t* cross,
                                   double neg_rate, uint32_t* mt_state,
                                   int32_t* e_src, int32_t* e_dst, int32_t* e_type, int64_t e_cap, int64_t* n_edges,
                                   int32_t* in_piece_idx, int64_t* n_in_piece,
                                   int32_t* pair_graph, int32_t* pair_m, int32_t* pair_n, int32_t* pair_gold,
                                   int64_t pair_cap, int64_t* n_pairs) {
    if (!n_atoms || !ad || !cross || !mt_state || !n_edges || !n_in_piece || !n_pairs) return PSVAE_E_NULL;
    if (B < 0 || N < 0) return PSVAE_E_SHAPE;
    int64_t ne = 0, ni = 0, np = 0, base = 0;
    for (int i = 0; i < B; ++i) {
        const int n = n_atoms[i];
        if (n < 0 || n > N) return PSVAE_E_SHAPE;
        const int8_t* A = ad + base;
        const int8_t* C = cross + base;
        const int32_t off = N * i;
        for (int m = 0; m < n; ++m) {
            for (int k = 0; k < n; ++k) {
                const int t = A[(int64_t)m * n + k];
                const int c = C[(int64_t)m * n + k];
                if (t != 0) {
                    if (ne >= e_cap) return PSVAE_E_CAPACITY;
                    if (c == 0) in_piece_idx[ni++] = (int32_t)ne;
                    e_src[ne] = m + off; e_dst[ne] = k + off; e_type[ne] = t;
                    ++ne;
                }
                if (c == 1 && (t != 0 || mt_random(mt_state) < neg_rate)) {
                    if (np >= pair_cap) return PSVAE_E_CAPACITY;
                    pair_graph[np] = i; pair_m[np] = m; pair_n[np] = k; pair_gold[np] = t;
                    ++np;
                }
            }
        }
        base += (int64_t)n * n;
    }
    *n_edges = ne; *n_in_piece = ni; *n_pairs = np;
    return 0;
}

// Stage 2 of the dataset (src/data/bpe_dataset.py:56-84) for the molecules `idx[0..B)` of a flattened
// data set, straight into the padded arrays and the concatenated matrices stage 3 walks.
extern "C" int psvae_gather_molecules(int B, const int64_t* idx, const int64_t* atom_off, const int32_t* x_all,
                                      const int32_t* piece_of_atom, const int32_t* pos_of_atom, const int32_t* group_of_atom,
                                      const int64_t* bond_off,
                                      const int32_t* bond_a, const int32_t* bond_b, const int8_t* bond_t,
                                      const int64_t* piece_off, const int32_t* pieces_all, int N, int L, int atom_pad,
                                      int piece_pad, int32_t* n_atoms, int32_t* x, int32_t* x_pieces, int32_t* x_pos,
                                      int32_t* pieces, int8_t* ad, int8_t* cross, int64_t matrix_bytes) {
    if (!idx || !atom_off || !x_all || !piece_of_atom || !pos_of_atom || !group_of_atom || !bond_off || !piece_off || !pieces_all ||
        !n_atoms || !x || !x_pieces || !x_pos || !pieces || !ad || !cross) return PSVAE_E_NULL;
    if (B < 1 || N < 1 || L < 1) return PSVAE_E_SHAPE;
    int64_t base = 0;
    for (int i = 0; i < B; ++i) {
        const int64_t m = idx[i], a0 = atom_off[m];
        const int64_t n = atom_off[m + 1] - a0, np = piece_off[m + 1] - piece_off[m];
        if (n < 0 || n > N || np < 0 || np > L) return PSVAE_E_SHAPE;
        if (base + n * n > matrix_bytes) return PSVAE_E_CAPACITY;
        n_atoms[i] = (int32_t)n;
        for (int64_t k = 0; k < N; ++k) {
            const bool real = k < n;
            x[(int64_t)i * N + k] = real ? x_all[a0 + k] : atom_pad;
            x_pieces[(int64_t)i * N + k] = real ? piece_of_atom[a0 + k] : piece_pad;
            x_pos[(int64_t)i * N + k] = real ? pos_of_atom[a0 + k] : 0;
        }
        for (int64_t k = 0; k < L; ++k) pieces[(int64_t)i * L + k] = k < np ? pieces_all[piece_off[m] + k] : piece_pad;
        int8_t* A = ad + base;
        int8_t* C = cross + base;
        memset(A, 0, (size_t)(n * n));
        for (int64_t e = bond_off[m]; e < bond_off[m + 1]; ++e) {      // list order: a later duplicate overwrites (:62-66)
            const int32_t u = bond_a[e], v = bond_b[e];
            if (u < 0 || v < 0 || u >= n || v >= n) return PSVAE_E_SHAPE;
            A[(int64_t)u * n + v] = bond_t[e];
            A[(int64_t)v * n + u] = bond_t[e];
        }
        // cross-piece mask (:68-82): 1 unless both atoms lie in the same group; an atom of no group (-1)
        // keeps its row of ones
        for (int64_t u = 0; u < n; ++u) {
            const int32_t gu = group_of_atom[a0 + u];
            for (int64_t v = 0; v < n; ++v) C[u * n + v] = (gu < 0 || gu != group_of_atom[a0 + v]) ? 1 : 0;
        }
        base += n * n;
    }
    return 0;
}

extern "C" int64_t psvae_pack_bound(int B, int N, int64_t n_edges, int64_t n_pairs, int L, int num_atom, int num_piece,
                                    int max_pos) {
    const int64_t n = (int64_t)B * N;
    return 8 * n + 2 * (B + 1) + 4 * (n + 1) + 12 * n_edges + 6 * n_pairs + 2 * (n + 1) + (int64_t)B * L * 2 +
           (num_atom + 1) + 2 * (num_piece + 1) + (max_pos + 1) + 64 * PSVAE_BLOB_FIELDS;
}

extern "C" int psvae_pack_batch(int B, int N, int L, const int32_t* n_atoms, const int32_t* x, const int32_t* x_pieces,
                                const int32_t* x_pos, const int32_t* e_src, const int32_t* e_dst, const int32_t* e_type,
                                int64_t n_edges, const int32_t* in_piece_idx, int64_t n_in_piece,
                                const int32_t* pair_graph, const int32_t* pair_m, const int32_t* pair_n,
                                const int32_t* pair_gold, int64_t n_pairs, const int32_t* pieces, int num_atom,
                                int num_piece, int max_pos, int pad_idx, int32_t* blob, int64_t blob_cap,
                                int64_t* offsets, int64_t* blob_words, int32_t* n_nodes_out, int32_t* n_target_tokens_out) {
    if (!n_atoms || !x || !x_pieces || !x_pos || !pieces || !blob || !offsets || !blob_words) return PSVAE_E_NULL;
    if (B < 1 || N < 1 || L < 2 || n_edges < 0 || n_in_piece < 0 || n_pairs < 0) return PSVAE_E_SHAPE;
    if (blob_cap < psvae_pack_bound(B, N, n_edges, n_pairs, L, num_atom, num_piece, max_pos)) return PSVAE_E_CAPACITY;
    std::vector<int32_t> goff(B + 1, 0);
    for (int i = 0; i < B; ++i) {
        if (n_atoms[i] < 0 || n_atoms[i] > N) return PSVAE_E_SHAPE;
        goff[i + 1] = goff[i] + n_atoms[i];
    }
    const int n = goff[B];
    const int T = L - 1;
    int64_t used = 0;
    auto field = [&](int id, int64_t count) -> int32_t* {
        offsets[id] = used;
        int32_t* p = blob + used;
        used += (count + 15) / 16 * 16 + 16;  // every field starts 64-byte aligned and is never empty
        return p;
    };
    // padded index -> packed row; fails on an index that points at a padding slot
    auto packed = [&](int32_t q, int32_t* out) -> bool {
        if (q < 0 || q >= B * N) return false;
        const int b = q / N, a = q - b * N;
        if (a >= n_atoms[b]) return false;
        *out = goff[b] + a;
        return true;
    };
    int32_t* atom = field(PSVAE_BLOB_ATOM_ID, n);
    int32_t* piece = field(PSVAE_BLOB_PIECE_ID, n);
    int32_t* pos = field(PSVAE_BLOB_POS_ID, n);
    int32_t* ngraph = field(PSVAE_BLOB_NODE_GRAPH, n);
    int32_t* goff_out = field(PSVAE_BLOB_GRAPH_OFF, B + 1);
    std::vector<int32_t> k_atom(n), k_piece(n), k_pos(n);
    for (int b = 0; b < B; ++b)
        for (int a = 0; a < n_atoms[b]; ++a) {
            const int r = goff[b] + a;
            const int64_t q = (int64_t)b * N + a;
            atom[r] = x[q]; piece[r] = x_pieces[q]; pos[r] = x_pos[q]; ngraph[r] = b;
            if (x[q] < 0 || x[q] >= num_atom || x_pieces[q] < 0 || x_pieces[q] >= num_piece || x_pos[q] < 0 || x_pos[q] >= max_pos)
                return PSVAE_E_SHAPE;
            k_atom[r] = x[q]; k_piece[r] = x_pieces[q]; k_pos[r] = x_pos[q];
        }
    memcpy(goff_out, goff.data(), sizeof(int32_t) * (B + 1));

    // bond graphs
    std::vector<int32_t> ps(n_edges), pd(n_edges);
    for (int64_t e = 0; e < n_edges; ++e) {
        if (!packed(e_src[e], &ps[e]) || !packed(e_dst[e], &pd[e])) return PSVAE_E_SHAPE;
        if (e_type[e] < 0 || e_type[e] > 3) return PSVAE_E_SHAPE;
    }
    auto build_csr = [&](const std::vector<int32_t>& rows, const std::vector<int32_t>& cols, const std::vector<int32_t>& types,
                         int id_ptr, int id_col, int id_type) {
        const int64_t m = (int64_t)rows.size();
        int32_t* ptr = field(id_ptr, n + 1);
        int32_t* col = field(id_col, m);
        int32_t* typ = field(id_type, m);
        std::vector<int32_t> order(m);
        bucket(rows, n, ptr, order.data());
        for (int64_t i = 0; i < m; ++i) { col[i] = cols[order[i]]; typ[i] = types[order[i]]; }
    };
    std::vector<int32_t> ty(e_type, e_type + n_edges);
    build_csr(pd, ps, ty, PSVAE_BLOB_ROWPTR, PSVAE_BLOB_COL, PSVAE_BLOB_ETYPE);
    build_csr(ps, pd, ty, PSVAE_BLOB_ROWPTR_T, PSVAE_BLOB_COL_T, PSVAE_BLOB_ETYPE_T);
    std::vector<int32_t> is(n_in_piece), idd(n_in_piece), it(n_in_piece);
    for (int64_t i = 0; i < n_in_piece; ++i) {
        const int32_t e = in_piece_idx[i];
        if (e < 0 || e >= n_edges) return PSVAE_E_SHAPE;
        is[i] = ps[e]; idd[i] = pd[e]; it[i] = ty[e];
    }
    build_csr(idd, is, it, PSVAE_BLOB_IN_ROWPTR, PSVAE_BLOB_IN_COL, PSVAE_BLOB_IN_ETYPE);
    build_csr(is, idd, it, PSVAE_BLOB_IN_ROWPTR_T, PSVAE_BLOB_IN_COL_T, PSVAE_BLOB_IN_ETYPE_T);

    // selected pairs
    int32_t* psrc = field(PSVAE_BLOB_PAIR_SRC, n_pairs);
    int32_t* pdst = field(PSVAE_BLOB_PAIR_DST, n_pairs);
    int32_t* pgr = field(PSVAE_BLOB_PAIR_GRAPH, n_pairs);
    int32_t* pgold = field(PSVAE_BLOB_PAIR_GOLD, n_pairs);
    std::vector<int32_t> ksrc(n_pairs), kdst(n_pairs), kgr(n_pairs);
    for (int64_t i = 0; i < n_pairs; ++i) {
        const int b = pair_graph[i];
        if (b < 0 || b >= B || (i > 0 && b < pair_graph[i - 1])) return PSVAE_E_SHAPE;
        if (pair_m[i] < 0 || pair_m[i] >= n_atoms[b] || pair_n[i] < 0 || pair_n[i] >= n_atoms[b]) return PSVAE_E_SHAPE;
        if (pair_gold[i] < 0 || pair_gold[i] > 3) return PSVAE_E_SHAPE;
        psrc[i] = ksrc[i] = goff[b] + pair_m[i];
        pdst[i] = kdst[i] = goff[b] + pair_n[i];
        pgr[i] = kgr[i] = b;
        pgold[i] = pair_gold[i];
    }
    {
        int32_t* ptr = field(PSVAE_BLOB_PAIR_BY_SRC_PTR, n + 1);
        int32_t* idx = field(PSVAE_BLOB_PAIR_BY_SRC_IDX, n_pairs);
        bucket(ksrc, n, ptr, idx);
    }
    {
        int32_t* ptr = field(PSVAE_BLOB_PAIR_BY_DST_PTR, n + 1);
        int32_t* idx = field(PSVAE_BLOB_PAIR_BY_DST_IDX, n_pairs);
        bucket(kdst, n, ptr, idx);
    }
    {
        int32_t* ptr = field(PSVAE_BLOB_PAIR_GRAPH_OFF, B + 1);
        std::vector<int32_t> order(n_pairs);
        bucket(kgr, B, ptr, order.data());
    }
    // piece sequences
    int32_t* pc = field(PSVAE_BLOB_PIECES, (int64_t)B * L);
    int32_t ntok = 0;
    std::vector<int32_t> kseq((size_t)B * T);
    for (int b = 0; b < B; ++b)
        for (int t = 0; t < L; ++t) {
            const int32_t v = pieces[(int64_t)b * L + t];
            if (v < 0 || v >= num_piece) return PSVAE_E_SHAPE;
            pc[(int64_t)b * L + t] = v;
            if (t < T) kseq[(size_t)b * T + t] = v;
            if (t >= 1 && v != pad_idx) ++ntok;
        }
    {
        int32_t* ptr = field(PSVAE_BLOB_ATOM_ROWS_PTR, num_atom + 1);
        int32_t* idx = field(PSVAE_BLOB_ATOM_ROWS_IDX, n);
        bucket(k_atom, num_atom, ptr, idx);
    }
    {
        int32_t* ptr = field(PSVAE_BLOB_PIECE_ROWS_PTR, num_piece + 1);
        int32_t* idx = field(PSVAE_BLOB_PIECE_ROWS_IDX, n);
        bucket(k_piece, num_piece, ptr, idx);
    }
    {
        int32_t* ptr = field(PSVAE_BLOB_POS_ROWS_PTR, max_pos + 1);
        int32_t* idx = field(PSVAE_BLOB_POS_ROWS_IDX, n);
        bucket(k_pos, max_pos, ptr, idx);
    }
    {
        int32_t* ptr = field(PSVAE_BLOB_SEQ_ROWS_PTR, num_piece + 1);
        int32_t* idx = field(PSVAE_BLOB_SEQ_ROWS_IDX, (int64_t)B * T);
        bucket(kseq, num_piece, ptr, idx);
    }
    if (used > blob_cap) return PSVAE_E_CAPACITY;
    *blob_words = used;
    if (n_nodes_out) *n_nodes_out = n;
    if (n_target_tokens_out) *n_target_tokens_out = ntok;
    return 0;
}
