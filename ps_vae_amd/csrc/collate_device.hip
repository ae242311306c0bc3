// Batch builder on the GPU: stages 2 and 3 of the reference's dataset (src/data/bpe_dataset.py:56-155) for a
// data set that already lies flattened in device memory, straight into the index blob the step kernels read.
//
// What a batch needs from a molecule is almost all static: its atoms, its bonds as a CSR in local atom numbers
// (for an undirected molecule the CSR by target and the CSR by source are the same arrays), the same for its
// in-piece bonds, its piece sequence.  The store keeps those per molecule; building a batch is then a copy with
// an atom offset and an edge offset added.  The one random part is the choice of negative pairs
// (bpe_dataset.py:129, `random() < 0.022` for every ordered cross-piece pair without a bond): here every
// candidate draws from a counter-based generator keyed by (seed, epoch, molecule, pair), so the choice does not
// depend on the batch a molecule lands in, on the position in it, or on the number of GPUs - and therefore is
// not the reference's MT19937 stream (the host collate reproduces that one bit for bit).
//
// All lists come out in the order the host builder produces (pairs row-major per graph, inverse lists stable),
// so a batch built here and one built on the host from the same selected pairs are the same blob.
#include "psvae_internal.h"

namespace {

constexpr int CD_NMAX = 128;    // atoms per molecule: the pair kernels keep a molecule's selected pairs as a bit matrix in LDS
constexpr int CD_TILE = 256;    // items per block of the stable counting sorts
constexpr int CD_MAXKEYS = 8192;

struct SortDesc { const int32_t* keys; int32_t *ptr, *idx, *tilecnt; int count, nkeys; };

struct CdArgs {
    psvae_mol_store st;
    const int32_t* mol;
    int B, n, L, T, pad_idx, n_prop_cols, prop_cols[8];
    long long E, E_in, S_cap;
    double rate;
    unsigned long long key0;
    int32_t* f[PSVAE_BLOB_FIELDS];
    int32_t *eoff, *einoff, *pcount, *seqkey, *counts;
    float *props_out, *denoms_out;
    SortDesc sort[4];
};

__device__ __forceinline__ unsigned long long mix64(unsigned long long z) {
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
// uniform in [0, 1) of ordered pair `pair` = m * n_atoms + k of the molecule whose key is `key_mol`
__device__ __forceinline__ double pair_uniform(unsigned long long key_mol, unsigned pair) {
    return (double)(mix64(key_mol ^ (unsigned long long)pair) >> 11) * (1.0 / 9007199254740992.0);
}

// exclusive scan of v over the 1024 threads of a block; returns the prefix, *total receives the sum
__device__ int block_scan_1024(int v, int* lds, int* total) {
    const int t = threadIdx.x;
    lds[t] = v;
    __syncthreads();
    for (int d = 1; d < 1024; d <<= 1) {
        const int add = t >= d ? lds[t - d] : 0;
        __syncthreads();
        lds[t] += add;
        __syncthreads();
    }
    const int incl = lds[t];
    *total = lds[1023];
    __syncthreads();
    return incl - v;
}

// graph_off, edge offsets, target-token count, longest sequence; checks the sizes the host passed
__global__ __launch_bounds__(1024) void cd_offsets_kernel(const CdArgs a) {
    __shared__ int lds[1024];
    __shared__ int red[4];
    const int t = threadIdx.x, per = (a.B + 1023) / 1024;
    const int b0 = min(a.B, t * per), b1 = min(a.B, b0 + per);
    int sa = 0, se = 0, si = 0, tok = 0, lmax = 0, bad = 0;
    for (int b = b0; b < b1; ++b) {
        const int m = a.mol[b];
        if (m < 0 || m >= a.st.n_mols) { bad = 1; continue; }
        const int na = a.st.atom_off[m + 1] - a.st.atom_off[m], np = a.st.piece_off[m + 1] - a.st.piece_off[m];
        sa += na; se += a.st.edge_off[m + 1] - a.st.edge_off[m]; si += a.st.in_edge_off[m + 1] - a.st.in_edge_off[m];
        tok += max(np - 1, 0); lmax = max(lmax, np);
        if (na > CD_NMAX || na < 1) bad = 1;
    }
    int ta, te, ti;
    int pa = block_scan_1024(sa, lds, &ta), pe = block_scan_1024(se, lds, &te), pi = block_scan_1024(si, lds, &ti);
    for (int b = b0; b < b1; ++b) {
        const int m = a.mol[b];
        a.f[PSVAE_BLOB_GRAPH_OFF][b] = pa; a.eoff[b] = pe; a.einoff[b] = pi;
        if (m < 0 || m >= a.st.n_mols) continue;
        pa += a.st.atom_off[m + 1] - a.st.atom_off[m];
        pe += a.st.edge_off[m + 1] - a.st.edge_off[m];
        pi += a.st.in_edge_off[m + 1] - a.st.in_edge_off[m];
    }
    if (t == 0) { red[0] = 0; red[1] = 0; red[2] = 0; }
    __syncthreads();
    atomicAdd(&red[0], tok); atomicMax(&red[1], lmax); atomicOr(&red[2], bad);
    __syncthreads();
    if (t == 0) {
        a.f[PSVAE_BLOB_GRAPH_OFF][a.B] = ta; a.eoff[a.B] = te; a.einoff[a.B] = ti;
        a.counts[1] = red[0]; a.counts[2] = ta; a.counts[3] = te; a.counts[4] = ti; a.counts[5] = red[1];
        a.counts[6] = red[2] | (ta != a.n ? 2 : 0) | (te != a.E ? 4 : 0) | (ti != a.E_in ? 8 : 0) | (red[1] != a.L ? 16 : 0);
        // closing row bounds of the four bond CSRs
        a.f[PSVAE_BLOB_ROWPTR][a.n] = (int)a.E; a.f[PSVAE_BLOB_ROWPTR_T][a.n] = (int)a.E;
        a.f[PSVAE_BLOB_IN_ROWPTR][a.n] = (int)a.E_in; a.f[PSVAE_BLOB_IN_ROWPTR_T][a.n] = (int)a.E_in;
    }
}

// one block per graph: atoms, both bond graphs, the piece sequence, the property targets
__global__ __launch_bounds__(256) void cd_graph_kernel(const CdArgs a) {
    const int i = blockIdx.x, tid = threadIdx.x, m = a.mol[i];
    if (m < 0 || m >= a.st.n_mols) return;
    const int a0 = a.st.atom_off[m], na = a.st.atom_off[m + 1] - a0, g0 = a.f[PSVAE_BLOB_GRAPH_OFF][i];
    const int e0 = a.st.edge_off[m], ne = a.st.edge_off[m + 1] - e0, eo = a.eoff[i];
    const int i0 = a.st.in_edge_off[m], ni = a.st.in_edge_off[m + 1] - i0, io = a.einoff[i];
    const int32_t* rp = a.st.rowptr_loc + a0 + m;
    const int32_t* irp = a.st.in_rowptr_loc + a0 + m;
    if (g0 + na > a.n || eo + ne > a.E || io + ni > a.E_in) return;   // sizes disagree with the host's: flagged by cd_offsets_kernel
    for (int k = tid; k < na; k += 256) {
        const int r = g0 + k;
        a.f[PSVAE_BLOB_ATOM_ID][r] = a.st.atom_id[a0 + k];
        a.f[PSVAE_BLOB_PIECE_ID][r] = a.st.piece_of_atom[a0 + k];
        a.f[PSVAE_BLOB_POS_ID][r] = a.st.pos_of_atom[a0 + k];
        a.f[PSVAE_BLOB_NODE_GRAPH][r] = i;
        a.f[PSVAE_BLOB_ROWPTR][r] = eo + rp[k]; a.f[PSVAE_BLOB_ROWPTR_T][r] = eo + rp[k];
        a.f[PSVAE_BLOB_IN_ROWPTR][r] = io + irp[k]; a.f[PSVAE_BLOB_IN_ROWPTR_T][r] = io + irp[k];
    }
    for (int e = tid; e < ne; e += 256) {
        const int c = g0 + a.st.nbr[e0 + e], ty = a.st.nbr_type[e0 + e];
        a.f[PSVAE_BLOB_COL][eo + e] = c; a.f[PSVAE_BLOB_COL_T][eo + e] = c;
        a.f[PSVAE_BLOB_ETYPE][eo + e] = ty; a.f[PSVAE_BLOB_ETYPE_T][eo + e] = ty;
    }
    for (int e = tid; e < ni; e += 256) {
        const int c = g0 + a.st.in_nbr[i0 + e], ty = a.st.in_nbr_type[i0 + e];
        a.f[PSVAE_BLOB_IN_COL][io + e] = c; a.f[PSVAE_BLOB_IN_COL_T][io + e] = c;
        a.f[PSVAE_BLOB_IN_ETYPE][io + e] = ty; a.f[PSVAE_BLOB_IN_ETYPE_T][io + e] = ty;
    }
    const int p0 = a.st.piece_off[m], np = a.st.piece_off[m + 1] - p0;
    for (int t = tid; t < a.L; t += 256) {
        const int v = t < np ? a.st.pieces[p0 + t] : a.pad_idx;
        a.f[PSVAE_BLOB_PIECES][(long long)i * a.L + t] = v;
        if (t < a.T) a.seqkey[(long long)i * a.T + t] = v;
    }
    if (tid < a.n_prop_cols) a.props_out[(long long)i * a.n_prop_cols + tid] = a.st.props[(long long)m * a.st.n_props_all + a.prop_cols[tid]];
}

// bond type of the ordered pair (u, v) of a molecule: 0 when there is no bond
__device__ __forceinline__ int bond_type(const int32_t* rp, const int32_t* nbr, const int32_t* typ, int u, int v) {
    int t = 0;
    for (int e = rp[u]; e < rp[u + 1]; ++e) t = nbr[e] == v ? typ[e] : t;
    return t;
}

// The selected ordered pairs (u, v) of graph i as a bit matrix in LDS (row u, bit v): cross-piece
// (bpe_dataset.py:68-82: different groups, or u in no group) and either bonded or drawn with probability `rate`.
__device__ void cd_select_pairs(const CdArgs& a, int m, int na, unsigned (*bits)[CD_NMAX / 32]) {
    const int tid = threadIdx.x, a0 = a.st.atom_off[m];
    for (int k = tid; k < CD_NMAX * (CD_NMAX / 32); k += blockDim.x) (&bits[0][0])[k] = 0u;
    __syncthreads();
    const int32_t* rp = a.st.rowptr_loc + a0 + m;
    const int32_t* nbr = a.st.nbr + a.st.edge_off[m];
    const int32_t* typ = a.st.nbr_type + a.st.edge_off[m];
    const int32_t* grp = a.st.group_of_atom + a0;
    const unsigned long long key_mol = mix64(a.key0 ^ (unsigned long long)(unsigned)m);
    for (int c = tid; c < na * na; c += blockDim.x) {
        const int u = c / na, v = c - u * na;
        const int gu = grp[u];
        if (!(gu < 0 || gu != grp[v])) continue;
        const bool keep = bond_type(rp, nbr, typ, u, v) != 0 || pair_uniform(key_mol, (unsigned)c) < a.rate;
        if (keep) atomicOr(&bits[u][v >> 5], 1u << (v & 31));
    }
    __syncthreads();
}

__global__ __launch_bounds__(256) void cd_pairs_count_kernel(const CdArgs a) {
    __shared__ unsigned bits[CD_NMAX][CD_NMAX / 32];
    __shared__ int total;
    const int i = blockIdx.x, tid = threadIdx.x, m = a.mol[i];
    if (tid == 0) total = 0;
    if (m < 0 || m >= a.st.n_mols) { if (tid == 0) a.pcount[i] = 0; return; }
    const int na = min(a.st.atom_off[m + 1] - a.st.atom_off[m], CD_NMAX);
    cd_select_pairs(a, m, na, bits);
    if (tid < na) {
        int c = 0;
        for (int w = 0; w < CD_NMAX / 32; ++w) c += __popc(bits[tid][w]);
        atomicAdd(&total, c);
    }
    __syncthreads();
    if (tid == 0) a.pcount[i] = total;
}

// pair_graph_off = exclusive scan of the per-graph counts; S; the loss denominators
__global__ __launch_bounds__(1024) void cd_pairs_scan_kernel(const CdArgs a) {
    __shared__ int lds[1024];
    const int t = threadIdx.x, per = (a.B + 1023) / 1024;
    const int b0 = min(a.B, t * per), b1 = min(a.B, b0 + per);
    int s = 0;
    for (int b = b0; b < b1; ++b) s += a.pcount[b];
    int S;
    int p = block_scan_1024(s, lds, &S);
    for (int b = b0; b < b1; ++b) { a.f[PSVAE_BLOB_PAIR_GRAPH_OFF][b] = p; p += a.pcount[b]; }
    if (t == 0) {
        a.f[PSVAE_BLOB_PAIR_GRAPH_OFF][a.B] = S;
        a.counts[0] = S;
        if (S > a.S_cap) a.counts[6] |= 32;
        a.f[PSVAE_BLOB_PAIR_BY_SRC_PTR][a.n] = S; a.f[PSVAE_BLOB_PAIR_BY_DST_PTR][a.n] = S;
        a.denoms_out[0] = (float)a.counts[1]; a.denoms_out[1] = (float)S; a.denoms_out[2] = (float)a.B;
        a.denoms_out[3] = (float)(a.B * a.n_prop_cols);
    }
}

// the pair lists of graph i: pairs in row-major order, grouped by source (that is the same order) and by target
__global__ __launch_bounds__(256) void cd_pairs_write_kernel(const CdArgs a) {
    __shared__ unsigned bits[CD_NMAX][CD_NMAX / 32];
    __shared__ int rowstart[CD_NMAX + 1], colstart[CD_NMAX + 1];
    const int i = blockIdx.x, tid = threadIdx.x, m = a.mol[i];
    if (m < 0 || m >= a.st.n_mols) return;
    const int a0 = a.st.atom_off[m], na = min(a.st.atom_off[m + 1] - a0, CD_NMAX), g0 = a.f[PSVAE_BLOB_GRAPH_OFF][i];
    const int P0 = a.f[PSVAE_BLOB_PAIR_GRAPH_OFF][i], P1 = a.f[PSVAE_BLOB_PAIR_GRAPH_OFF][i + 1];
    if (P1 > a.S_cap || g0 + na > a.n) return;     // flagged elsewhere
    cd_select_pairs(a, m, na, bits);
    if (tid < na) {
        int c = 0;
        for (int w = 0; w < CD_NMAX / 32; ++w) c += __popc(bits[tid][w]);
        rowstart[tid + 1] = c;
        int d = 0;
        for (int u = 0; u < na; ++u) d += (bits[u][tid >> 5] >> (tid & 31)) & 1u;
        colstart[tid + 1] = d;
    }
    __syncthreads();
    if (tid == 0) {
        rowstart[0] = 0; colstart[0] = 0;
        for (int u = 0; u < na; ++u) { rowstart[u + 1] += rowstart[u]; colstart[u + 1] += colstart[u]; }
    }
    __syncthreads();
    const int32_t* rp = a.st.rowptr_loc + a0 + m;
    const int32_t* nbr = a.st.nbr + a.st.edge_off[m];
    const int32_t* typ = a.st.nbr_type + a.st.edge_off[m];
    auto rank_in_row = [&](int u, int v) {     // selected pairs (u, v') with v' < v
        int r = 0;
        for (int w = 0; w < (v >> 5); ++w) r += __popc(bits[u][w]);
        return r + __popc(bits[u][v >> 5] & ((1u << (v & 31)) - 1u));
    };
    for (int c = tid; c < na * na; c += 256) {
        const int u = c / na, v = c - u * na;
        if (!((bits[u][v >> 5] >> (v & 31)) & 1u)) continue;
        const int p = P0 + rowstart[u] + rank_in_row(u, v);
        a.f[PSVAE_BLOB_PAIR_SRC][p] = g0 + u; a.f[PSVAE_BLOB_PAIR_DST][p] = g0 + v;
        a.f[PSVAE_BLOB_PAIR_GRAPH][p] = i; a.f[PSVAE_BLOB_PAIR_GOLD][p] = bond_type(rp, nbr, typ, u, v);
        a.f[PSVAE_BLOB_PAIR_BY_SRC_IDX][p] = p;
    }
    if (tid < na) {
        a.f[PSVAE_BLOB_PAIR_BY_SRC_PTR][g0 + tid] = P0 + rowstart[tid];
        a.f[PSVAE_BLOB_PAIR_BY_DST_PTR][g0 + tid] = P0 + colstart[tid];
        int q = P0 + colstart[tid];
        for (int u = 0; u < na; ++u)
            if ((bits[u][tid >> 5] >> (tid & 31)) & 1u) a.f[PSVAE_BLOB_PAIR_BY_DST_IDX][q++] = P0 + rowstart[u] + rank_in_row(u, tid);
    }
}

// ---- stable counting sort of up to four key arrays (blockIdx.y = which): items grouped by key, in item order
__global__ __launch_bounds__(CD_TILE) void cd_sort_hist_kernel(const CdArgs a) {
    extern __shared__ int hist[];
    const SortDesc& s = a.sort[blockIdx.y];
    const int tile = blockIdx.x, tid = threadIdx.x, item = tile * CD_TILE + tid;
    if (tile * CD_TILE >= s.count) return;
    for (int k = tid; k < s.nkeys; k += CD_TILE) hist[k] = 0;
    __syncthreads();
    if (item < s.count) {
        const int key = s.keys[item];
        if (key >= 0 && key < s.nkeys) atomicAdd(&hist[key], 1); else atomicOr(&a.counts[6], 64);
    }
    __syncthreads();
    for (int k = tid; k < s.nkeys; k += CD_TILE) s.tilecnt[(long long)tile * s.nkeys + k] = hist[k];
}

// ptr[key] and, in place of each tile's count, the position its first item of that key goes to
__global__ __launch_bounds__(1024) void cd_sort_scan_kernel(const CdArgs a) {
    __shared__ int lds[1024];
    const SortDesc& s = a.sort[blockIdx.x];
    const int t = threadIdx.x, ntiles = (s.count + CD_TILE - 1) / CD_TILE;
    int carry = 0;
    for (int k0 = 0; k0 < s.nkeys; k0 += 1024) {     // uniform trip count over the block
        const int key = k0 + t;
        int tot = 0;
        if (key < s.nkeys)
            for (int tile = 0; tile < ntiles; ++tile) tot += s.tilecnt[(long long)tile * s.nkeys + key];
        int chunk;
        int run = carry + block_scan_1024(tot, lds, &chunk);
        carry += chunk;
        if (key < s.nkeys) {
            s.ptr[key] = run;
            for (int tile = 0; tile < ntiles; ++tile) {
                const long long q = (long long)tile * s.nkeys + key;
                const int c = s.tilecnt[q];
                s.tilecnt[q] = run;
                run += c;
            }
        }
    }
    if (t == 0) s.ptr[s.nkeys] = s.count;
}

__global__ __launch_bounds__(CD_TILE) void cd_sort_place_kernel(const CdArgs a) {
    __shared__ int keys[CD_TILE];
    const SortDesc& s = a.sort[blockIdx.y];
    const int tile = blockIdx.x, tid = threadIdx.x, item = tile * CD_TILE + tid;
    if (tile * CD_TILE >= s.count) return;
    const int key = item < s.count ? s.keys[item] : -1;
    keys[tid] = key;
    __syncthreads();
    if (key < 0 || key >= s.nkeys) return;
    int rank = 0;
    for (int j = 0; j < tid; ++j) rank += keys[j] == key ? 1 : 0;
    s.idx[s.tilecnt[(long long)tile * s.nkeys + key] + rank] = item;
}

// every field starts 64-byte aligned and is never empty (the layout psvae_pack_batch uses)
int64_t cd_layout(int B, int n, int64_t E, int64_t E_in, int64_t S, int L, int num_atom, int num_piece, int max_pos, int64_t* off) {
    const int64_t T = L - 1;
    const int64_t cnt[PSVAE_BLOB_FIELDS] = {
        n, n, n, n, (int64_t)B + 1,
        (int64_t)n + 1, E, E, (int64_t)n + 1, E, E,
        (int64_t)n + 1, E_in, E_in, (int64_t)n + 1, E_in, E_in,
        S, S, S, S, (int64_t)n + 1, S, (int64_t)n + 1, S, (int64_t)B + 1, (int64_t)B * L,
        (int64_t)num_atom + 1, n, (int64_t)num_piece + 1, n, (int64_t)max_pos + 1, n, (int64_t)num_piece + 1, (int64_t)B * T};
    int64_t used = 0;
    for (int k = 0; k < PSVAE_BLOB_FIELDS; ++k) {
        if (off) off[k] = used;
        used += (cnt[k] + 15) / 16 * 16 + 16;
    }
    return used;
}

struct CdScratch { size_t eoff, einoff, pcount, seqkey, tilecnt[4], total; };
CdScratch cd_scratch(int B, int n, int L, int num_atom, int num_piece, int max_pos) {
    CdScratch s{};
    size_t used = 0;
    auto take = [&](size_t words) { const size_t at = used; used += (words + 15) / 16 * 16; return at; };
    const int64_t T = L - 1;
    s.eoff = take((size_t)B + 1); s.einoff = take((size_t)B + 1); s.pcount = take((size_t)B); s.seqkey = take((size_t)B * T);
    const int64_t counts[4] = {n, n, n, (int64_t)B * T};
    const int nkeys[4] = {num_atom, num_piece, max_pos, num_piece};
    for (int k = 0; k < 4; ++k) s.tilecnt[k] = take((size_t)((counts[k] + CD_TILE - 1) / CD_TILE) * nkeys[k]);
    s.total = used;
    return s;
}

}  // namespace

extern "C" int64_t psvae_collate_device_layout(int B, int n, int64_t E, int64_t E_in, int64_t S_cap, int L, int num_atom,
                                               int num_piece, int max_pos, int64_t* offsets) {
    if (B < 1 || n < 1 || E < 0 || E_in < 0 || S_cap < 0 || L < 2 || num_atom < 1 || num_piece < 1 || max_pos < 1) return -1;
    return cd_layout(B, n, E, E_in, S_cap, L, num_atom, num_piece, max_pos, offsets);
}

extern "C" size_t psvae_collate_device_scratch_bytes(int B, int n, int L, int num_atom, int num_piece, int max_pos) {
    if (B < 1 || n < 1 || L < 2 || num_atom < 1 || num_piece < 1 || max_pos < 1) return 0;
    return cd_scratch(B, n, L, num_atom, num_piece, max_pos).total * sizeof(int32_t);
}

extern "C" int psvae_collate_device(const psvae_mol_store* store, const int32_t* mol_idx, int B, int n, int64_t E, int64_t E_in,
                                    int L, int64_t S_cap, double neg_rate, uint64_t seed, uint64_t epoch, int num_atom,
                                    int num_piece, int max_pos, int pad_idx, const int32_t* prop_cols, int n_prop_cols,
                                    int32_t* blob, int64_t blob_words, float* props_out, float* denoms_out,
                                    int32_t* counts_out, void* scratch, size_t scratch_bytes, void* stream) {
    if (!store || !mol_idx || !blob || !props_out || !denoms_out || !counts_out || !scratch || !prop_cols) return PSVAE_E_NULL;
    if (!store->atom_off || !store->atom_id || !store->piece_of_atom || !store->pos_of_atom || !store->group_of_atom ||
        !store->edge_off || !store->rowptr_loc || !store->nbr || !store->nbr_type || !store->in_edge_off ||
        !store->in_rowptr_loc || !store->in_nbr || !store->in_nbr_type || !store->piece_off || !store->pieces || !store->props)
        return PSVAE_E_NULL;
    if (B < 1 || n < 1 || E < 0 || E_in < 0 || L < 2 || S_cap < 0 || n_prop_cols < 1 || n_prop_cols > 8) return PSVAE_E_SHAPE;
    if (num_atom < 1 || num_piece < 1 || max_pos < 1 || num_atom > CD_MAXKEYS || num_piece > CD_MAXKEYS || max_pos > CD_MAXKEYS)
        return PSVAE_E_SHAPE;
    if (E >= (1LL << 31) || S_cap >= (1LL << 31) || (int64_t)B * L >= (1LL << 31) || store->n_mols >= (1LL << 31)) return PSVAE_E_SHAPE;
    for (int k = 0; k < n_prop_cols; ++k)
        if (prop_cols[k] < 0 || prop_cols[k] >= store->n_props_all) return PSVAE_E_SHAPE;
    CdArgs a{};
    int64_t off[PSVAE_BLOB_FIELDS];
    if (cd_layout(B, n, E, E_in, S_cap, L, num_atom, num_piece, max_pos, off) > blob_words) return PSVAE_E_CAPACITY;
    const CdScratch sc = cd_scratch(B, n, L, num_atom, num_piece, max_pos);
    if (sc.total * sizeof(int32_t) > scratch_bytes) return PSVAE_E_WORKSPACE;
    a.st = *store; a.mol = mol_idx; a.B = B; a.n = n; a.L = L; a.T = L - 1; a.pad_idx = pad_idx;
    a.n_prop_cols = n_prop_cols;
    for (int k = 0; k < n_prop_cols; ++k) a.prop_cols[k] = prop_cols[k];
    a.E = E; a.E_in = E_in; a.S_cap = S_cap; a.rate = neg_rate;
    {   // key of (seed, epoch), mixed on the host exactly as the kernels mix the molecule into it
        auto mix = [](uint64_t z) { z += 0x9E3779B97F4A7C15ull; z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
                                    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; return z ^ (z >> 31); };
        a.key0 = mix(mix(seed) ^ epoch);
    }
    for (int k = 0; k < PSVAE_BLOB_FIELDS; ++k) a.f[k] = blob + off[k];
    int32_t* s32 = (int32_t*)scratch;
    a.eoff = s32 + sc.eoff; a.einoff = s32 + sc.einoff; a.pcount = s32 + sc.pcount; a.seqkey = s32 + sc.seqkey;
    a.counts = counts_out; a.props_out = props_out; a.denoms_out = denoms_out;
    const int T = L - 1;
    a.sort[0] = SortDesc{a.f[PSVAE_BLOB_ATOM_ID], a.f[PSVAE_BLOB_ATOM_ROWS_PTR], a.f[PSVAE_BLOB_ATOM_ROWS_IDX], s32 + sc.tilecnt[0], n, num_atom};
    a.sort[1] = SortDesc{a.f[PSVAE_BLOB_PIECE_ID], a.f[PSVAE_BLOB_PIECE_ROWS_PTR], a.f[PSVAE_BLOB_PIECE_ROWS_IDX], s32 + sc.tilecnt[1], n, num_piece};
    a.sort[2] = SortDesc{a.f[PSVAE_BLOB_POS_ID], a.f[PSVAE_BLOB_POS_ROWS_PTR], a.f[PSVAE_BLOB_POS_ROWS_IDX], s32 + sc.tilecnt[2], n, max_pos};
    a.sort[3] = SortDesc{a.seqkey, a.f[PSVAE_BLOB_SEQ_ROWS_PTR], a.f[PSVAE_BLOB_SEQ_ROWS_IDX], s32 + sc.tilecnt[3], B * T, num_piece};
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(cd_offsets_kernel, dim3(1), dim3(1024), 0, st, a);
    PSVAE_LAUNCH_CHECK();
    hipLaunchKernelGGL(cd_graph_kernel, dim3(B), dim3(256), 0, st, a);
    PSVAE_LAUNCH_CHECK();
    hipLaunchKernelGGL(cd_pairs_count_kernel, dim3(B), dim3(256), 0, st, a);
    PSVAE_LAUNCH_CHECK();
    hipLaunchKernelGGL(cd_pairs_scan_kernel, dim3(1), dim3(1024), 0, st, a);
    PSVAE_LAUNCH_CHECK();
    hipLaunchKernelGGL(cd_pairs_write_kernel, dim3(B), dim3(256), 0, st, a);
    PSVAE_LAUNCH_CHECK();
    const int maxcount = n > B * T ? n : B * T, maxkeys = num_piece > num_atom ? (num_piece > max_pos ? num_piece : max_pos) : (num_atom > max_pos ? num_atom : max_pos);
    const dim3 sgrid((maxcount + CD_TILE - 1) / CD_TILE, 4);
    hipLaunchKernelGGL(cd_sort_hist_kernel, sgrid, dim3(CD_TILE), (size_t)maxkeys * sizeof(int), st, a);
    PSVAE_LAUNCH_CHECK();
    hipLaunchKernelGGL(cd_sort_scan_kernel, dim3(4), dim3(1024), 0, st, a);
    PSVAE_LAUNCH_CHECK();
    hipLaunchKernelGGL(cd_sort_place_kernel, sgrid, dim3(CD_TILE), 0, st, a);
    PSVAE_LAUNCH_CHECK();
    return 0;
}
