// blocked.cpp — host-side driver of the blocked exact GP (blocked.h).  Plain C++: compiled by hipcc into libgpmi.so (back end:
// dev_hip.hip) and by g++ into the CPU test library (back end: tests/hostdev/host_dev.cpp).
//
// FACTORISATION (fit): right-looking over block columns of WD rows, the two-level scheme of chol.h with the super-panel as
// the distributed block.  Step k on every rank, with the gathered panel P_k (solved rows of block column k, global order):
//   U1   own rows below block k  x  block column k+1          -= X_k P_k[k+1]'      (narrow; makes column k+1 current)
//   chain (owner of block k+1, stream DS_SIDE)                 dpotrf of the diagonal block + its explicit inverse LW_{k+1}
//   bcast LW_{k+1} (DS_SIDE)
//   U2a  own rows  x  block columns [k+2, m)                   under the chain and the broadcast (half of what is left)
//   solve next panel: X_{k+1} <- X_{k+1} LW_{k+1}'  (out of place into S = the send buffer, copied back)
//   all-gather of S into P_{k+1} (DS_SIDE)                     under U2b
//   U2b  own rows  x  block columns [m, nblk)
// so the chain, the broadcast and the panel exchange are all off the critical path (one-step look-ahead, HPL style); the
// only host synchronisations are at the start and at the end of a call.  The right-hand side y - mu rides along as one extra
// ("carried") row on every rank: when the loop ends it holds z = L^-1 (y - mu).
#include "blocked.h"

#include <cstring>

#include <math.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>

namespace gpmi {

static const double LOG2PI = 1.8378770664093453;
// rows left below which a step of a ONE-rank factorisation reserves whole CUs for the chain (common.h whole_cus_below; with more ranks every
// step does).  20480 in rounds 3-5; since the one-rank update's grid covers every compute unit and the chain is placed first (round 6) free
// slots win at every size: 20480 / 12288 / 0 -> N = 50 000 709 / 708 / 703 ms, N = 20 000 69.8 / 68.8 / 69.0 (profiles/r06_p_*).
// GPMI_BLOCKED_WHOLE_BELOW: test / sweep hook, read once.
static int64_t whole_cus_below_rows() {
    static const int64_t v = [] {
        const char* e = getenv("GPMI_BLOCKED_WHOLE_BELOW");
        return e ? (int64_t)atoll(e) : (int64_t)0;
    }();
    return v;
}
#define kWholeCusBelow whole_cus_below_rows()

// 1024-row blocks once the K = 1024 update outlasts a 1024-block chain for most of the factorisation; narrower blocks keep the
// chain (and the exposed tail) short below that
static int64_t default_block(int64_t n, int world) {
    if (world == 1 && n >= 131072) return 2048;  // one rank, very large n: K = 2048 updates (N = 200 000 fp32: 20.9 instead of 22.0 s); below that the 2048-block chain outlasts the shrinking updates (N = 50 000: 790 instead of 760 ms, profiles/r03_h_*)
    return n >= 16384 ? 1024 : (n >= 4096 ? 512 : 256);  // (N = 20 000 on one rank: 82.9 ms with 1024-row blocks, 86.1 with 512: profiles/r04_g_*)
}

BlockedGP::BlockedGP(Dev* dev, Comm* comm, int d, int64_t n, BlockedOpts o)
    : dev_(dev), comm_(comm), rank_(comm ? comm->rank : 0), G_(comm ? comm->world : 1), d_(d), n_(n) {
    dev_->set_world(G_);
    es_ = dev->es;
    WD_ = o.block > 0 ? o.block : default_block(n, G_);
    nblk_ = (n + WD_ - 1) / WD_;
    npad_ = nblk_ * WD_;
    tpb_ = (int)(WD_ / 128);
    ldP_ = padded(WD_);
    for (int64_t b = rank_; b < nblk_; b += G_) own_.push_back((int)b);
    nown_ = (int)own_.size();
    maxown_ = (int)((nblk_ + G_ - 1) / G_);
    per_ = (o.stripe_blocks <= 0 || o.stripe_blocks >= nown_) ? std::max(nown_, 1) : o.stripe_blocks;
    u2a_cover_s_ = 1.1e-3 * ((double)WD_ / 1024.0) * ((double)WD_ / 1024.0) + 0.4e-3;  // chain kernel + inverse broadcast + margin
    if (const char* e = getenv("GPMI_BLOCKED_U2A_US")) u2a_cover_s_ = 1e-6 * atof(e);     // tuning / test hook
#ifdef GPMI_TOOLS  // tuning knob of the bring-up build (make TOOLS=1): U2a = 1 / u2a_div_ of the remaining block columns
    if (const char* e = getenv("GPMI_BLOCKED_U2A")) u2a_div_ = std::max(1, atoi(e));
#endif
}

BlockedGP::~BlockedGP() {
    for (void* p : allocs_) dev_->release(p);
    for (char* p : {xp_, Rloc_, Vk_, small_, Kpp_, G1_, Vb_, Wt_, (char*)dacc_, Bfull_, Bc_, (char*)dfull_})
        if (p) dev_->release(p);
}

void* BlockedGP::grab(int64_t bytes, bool zero) {
    void* p = dev_->alloc(std::max<int64_t>(bytes, 64));
    if (!p) return nullptr;
    allocs_.push_back(p);
    if (zero) dev_->zero(p, bytes);
    return p;
}

int BlockedGP::grow(char** p, int64_t* cap, int64_t bytes) {
    if (*cap >= bytes) return GPMI_OK;
    if (*p) dev_->release(*p);
    *cap = 0;
    *p = (char*)dev_->alloc(bytes);
    if (!*p) return fail(GPMI_EDEVICE, "blocked GP: out of device memory (" + dev_->err + ")");
    *cap = bytes;
    return GPMI_OK;
}

int BlockedGP::check_dev(const char* where) {
    if (!dev_->err.empty()) return fail(GPMI_EDEVICE, std::string(where) + ": " + dev_->err);
    if (comm_rc_) return fail(GPMI_EDEVICE, std::string(where) + ": a collective failed (rc " + std::to_string(comm_rc_) + ")");
    return GPMI_OK;
}

int BlockedGP::init(const void* x_host) {
    if (WD_ < 256 || WD_ % 256 || ((WD_ / 256) & (WD_ / 256 - 1))) return fail(GPMI_EARG, "the distributed block must be 256 * 2^s rows");
    dev_->begin_call();
    dev_->use(DS_MAIN);
    // stripes of `per_` local blocks; stripe s holds the columns up to the diagonal of its last block (the last stripe, which
    // also carries the y - mu row, all npad): N^2 (1 + 1/S) / 2 elements instead of N^2
    for (int i0 = 0; i0 < std::max(nown_, 1); i0 += per_) {
        const int i1 = std::min(i0 + per_, nown_);
        const bool last = i1 >= nown_;
        Stripe s;
        s.i0 = i0;
        s.i1 = std::max(i1, i0);
        s.width = (last || nown_ == 0) ? npad_ : (int64_t)(own_[i1 - 1] + 1) * WD_;
        s.ld = padded(s.width);
        s.rows = (int64_t)(s.i1 - s.i0) * WD_ + (last ? 8 : 0);
        s.p = (char*)grab(s.rows * s.ld * es_, true);
        if (!s.p) return fail(GPMI_EDEVICE, "blocked GP: out of device memory for the factor (" + dev_->err + ")");
        stored_bytes_ += s.rows * s.ld * es_;
        stripes_.push_back(s);
        if (last) break;
    }
    const int64_t srows = (int64_t)maxown_ * WD_ + 8;
    bool ok = true;
    auto take = [&](char** dst, int64_t bytes, bool zero) {
        *dst = (char*)grab(bytes, zero);
        ok = ok && *dst != nullptr;
    };
    take(&x_, n_ * d_ * es_, false);
    take(&LW_, nblk_ * WD_ * WD_ * es_, true);
    take(&linv_, (int64_t)std::max(nown_, 1) * WD_ * 64 * es_, true);
    take(&invd_, (int64_t)std::max(nown_, 1) * WD_ * es_, true);
    take(&alpha_, npad_ * es_, true);
    take(&v_, npad_ * es_, true);
    take(&ymu_, npad_ * es_, true);
    take(&seg_, WD_ * es_, true);
    take(&S_[0], srows * ldP_ * es_, true);
    if (G_ == 1) {
        take(&S_[1], srows * ldP_ * es_, true);  // one rank: the solved image IS the panel in global order; two of them alternate
    } else {
        const int64_t prow = (int64_t)maxown_ * G_ * WD_;  // whole groups of G blocks: the gather of the last group may run past npad
        take(&P_[0], prow * ldP_ * es_, true);
        take(&P_[1], prow * ldP_ * es_, true);
    }
    if (!ok) return fail(GPMI_EDEVICE, "blocked GP: out of device memory (" + dev_->err + ")");
    dev_->upload(x_, x_host, n_ * d_ * es_);
    dev_->sync();
    return check_dev("blocked GP init");
}

const BlockedGP::Stripe& BlockedGP::stripe_of(int i) const {
    for (const auto& s : stripes_)
        if (i >= s.i0 && i < s.i1) return s;
    return stripes_.back();
}
char* BlockedGP::block_ptr(int i, int64_t* ld, int64_t* width) const {
    const Stripe& s = stripe_of(i);
    *ld = s.ld;
    *width = s.width;
    return s.p + (int64_t)(i - s.i0) * WD_ * s.ld * es_;
}
char* BlockedGP::carried_ptr(int64_t* ld) const {
    const Stripe& s = stripes_.back();
    *ld = s.ld;
    return s.p + (int64_t)(s.i1 - s.i0) * WD_ * s.ld * es_;
}
// the local blocks >= first_block, stripe by stripe; the last piece ends with the carried row when asked for
std::vector<BlockedGP::Piece> BlockedGP::pieces(int first_block, bool carried) const {
    std::vector<Piece> out;
    for (size_t si = 0; si < stripes_.size(); ++si) {
        const Stripe& s = stripes_[si];
        const bool last = si + 1 == stripes_.size();
        const int b0 = std::max(s.i0, first_block);
        const int nb = std::max(0, s.i1 - b0);
        const bool extra = carried && last;
        if (nb == 0 && !extra) continue;
        Piece p;
        const int64_t r0 = nb ? (int64_t)(b0 - s.i0) * WD_ : (int64_t)(s.i1 - s.i0) * WD_;
        p.p = s.p + r0 * s.ld * es_;
        p.ld = s.ld;
        p.width = s.width;
        p.b0 = nb ? b0 : s.i1;
        p.nb = nb;
        p.carried = extra;
        out.push_back(p);
    }
    return out;
}
// rows of global block b (> k) of the gathered panel k
char* BlockedGP::panel_rows(int64_t k, int64_t b) const {
    if (G_ == 1) return S_[k & 1] + (b - k - 1) * WD_ * ldP_ * es_;
    return P_[k & 1] + b * WD_ * ldP_ * es_;
}

void BlockedGP::bcast_lw(int64_t k, DevEvent after) {
    if (G_ == 1) {  // one rank: the inverse is where it was built; whoever needs it waits for the chain
        ev_lw_ = after;
        return;
    }
    dev_->use(DS_COMM);
    dev_->wait(after);
    dev_->phase(GPMI_PROF_STEP_BCAST, true);
    comm_rc_ |= comm_->broadcast(LW_ + k * WD_ * WD_ * es_, WD_ * WD_ * es_, (int)(k % G_), dev_->native_stream());
    dev_->phase(GPMI_PROF_STEP_BCAST, false);
    ev_lw_ = dev_->record();
}

// Rows below block k (own blocks with global index > k, plus the carried row): X <- X LW_k' out of place into S (which is the
// send buffer of the exchange and, on one rank, the panel itself), copied back into the factor; then the all-gather into P_k
// in global row order.  from_factor: the rows are already solved (gradient: the panel is re-gathered from the stored factor).
void BlockedGP::solve_and_gather(int64_t k, bool from_factor) {
    const int64_t k0 = k * WD_;
    const int nle = n_le(rank_, k);
    char* S = S_[G_ == 1 ? (k & 1) : 0];
    dev_->use(DS_UPD);
    if (!from_factor) dev_->wait(ev_lw_);
    int64_t srow = 0;
    DevShape rect;
    if (!from_factor) dev_->phase(GPMI_PROF_STEP_SOLVE, true);
    for (const Piece& pc : pieces(nle, !from_factor)) {
        const int64_t M = (int64_t)pc.nb * WD_ + (pc.carried ? 1 : 0);
        char* X = pc.p + k0 * es_;
        char* Si = S + srow * ldP_ * es_;
        if (from_factor) {
            dev_->copy2d(Si, ldP_ * es_, X, pc.ld * es_, WD_ * es_, M);
        } else {
            dev_->gemm(Si, ldP_, X, pc.ld, LW_ + k * WD_ * WD_ * es_, WD_, M, WD_, WD_, rect, DG_OVERWRITE | DG_KEND_COL);
            dev_->copy2d(X, pc.ld * es_, Si, ldP_ * es_, WD_ * es_, M);
        }
        srow += M;
    }
    if (!from_factor) dev_->phase(GPMI_PROF_STEP_SOLVE, false);
    DevEvent ev_sr = dev_->record();
    ev_p_ = ev_sr;
    if (G_ == 1 || k + 1 >= nblk_) return;
    dev_->use(DS_COMM);
    dev_->wait(ev_sr);
    // straight into P_k in GLOBAL row order, no staging copy: block b lives on rank b mod G, so the G blocks [g G, (g + 1) G) are one
    // equal-size all-gather whose slot order IS the global order — one collective per group of G blocks (a communicator that can
    // fuses them: Comm::group_begin / group_end = ncclGroupStart / End).  A rank whose block of the group is not below k (the first
    // group) or past the end (the last) contributes a block of its send buffer that nobody reads.
    char* P = P_[k & 1];
    const int64_t blk = WD_ * ldP_ * es_;
    if (!from_factor) dev_->phase(GPMI_PROF_STEP_GATHER, true);
    comm_->group_begin();
    for (int64_t g = (k + 1) / G_; g * G_ < nblk_; ++g) {
        const int64_t b = g * G_ + rank_;  // my block of this group; its local index is g
        const char* send = (b > k && b < nblk_) ? S + (g - nle) * blk : S;
        comm_rc_ |= comm_->all_gather(send, P + g * G_ * blk, blk, dev_->native_stream());
    }
    comm_rc_ |= comm_->group_end();
    if (!from_factor) dev_->phase(GPMI_PROF_STEP_GATHER, false);
    ev_p_ = dev_->record();
}

void BlockedGP::join_on_main() {
    DevEvent ev[3];
    const DevStream ss[3] = {DS_SIDE, DS_COMM, DS_UPD};
    for (int i = 0; i < 3; ++i) {
        dev_->use(ss[i]);
        ev[i] = dev_->record();
    }
    dev_->use(DS_MAIN);
    for (int i = 0; i < 3; ++i) dev_->wait(ev[i]);
}

// own rows (and the carried row) x block columns [c_lo, c_hi) -= X_k P_k' : the staircase of a block-cyclic shard, one launch
// per stripe.  Only rows whose diagonal lies at or right of c_lo have entries there.
void BlockedGP::update_cols(int64_t k, int64_t c_lo, int64_t c_hi, int64_t min_block) {
    if (c_lo >= c_hi) return;
    const int first = n_le(rank_, std::max(c_lo, min_block) - 1);
    for (const Piece& pc : pieces(first, true)) {
        const int64_t M = (int64_t)pc.nb * WD_ + (pc.carried ? 1 : 0);
        const int64_t ncols = std::min<int64_t>(c_hi * WD_, pc.width) - c_lo * WD_;
        if (M <= 0 || ncols <= 0) continue;
        DevShape sh;
        if (G_ == 1) {
            // one rank: the shard IS the matrix — the plain lower region in tiles (column tile <= row tile + g0), which is the shape
            // the 256 x 128 update kernel takes (update256.hip); the staircase below would also keep the upper tiles of every
            // diagonal block
            sh.mode = 1;
            sh.g0 = pc.nb ? (int)(own_[pc.b0] - c_lo) * tpb_ : (int)(nblk_ - c_lo) * tpb_;
        } else {
            sh.mode = 2;
            sh.g0 = pc.nb ? (int)(own_[pc.b0] - c_lo) : 0;
            sh.G = G_;
            sh.nstair = tpb_ * pc.nb;
            sh.tpb = tpb_;
        }
        dev_->gemm(pc.p + c_lo * WD_ * es_, pc.ld, pc.p + k * WD_ * es_, pc.ld, panel_rows(k, c_lo), ldP_, M, ncols, WD_, sh, 0);
    }
}

int BlockedGP::fit(const gpmi_kernel* kern, const double* log_noise, int64_t n_noise, const void* ymu_host, double* mll_out, void* alpha_out,
                   int64_t* info_out) {
    fitted_ = false;
    comm_rc_ = 0;
    dev_->err.clear();
    if (info_out) *info_out = 0;
    if (n_noise != 1 && n_noise != n_) return fail(GPMI_EARG, "gpmi_fit: bad argument (logNoise must have length 1 or nobs)");
    dev_->begin_call();
    dev_->use(DS_MAIN);
    int n_hyp = 0;
    int rc = dev_->set_kernel(kern, d_, &kdiag_, &n_hyp);
    if (rc != GPMI_OK) return fail(rc, dev_->err);
    double nugget = 0.0;
    const double* nvec = nullptr;
    if (n_noise == 1) {
        nugget = exp(2.0 * log_noise[0]);  // GPE.jl:173
    } else {
        if (!noise_) noise_ = (double*)grab(n_ * 8, false);
        if (!noise_) return fail(GPMI_EDEVICE, "blocked GP: out of device memory");
        std::vector<double> nv((size_t)n_);
        for (int64_t i = 0; i < n_; ++i) nv[(size_t)i] = exp(2.0 * log_noise[i]);  // GPE.jl:181-183
        dev_->upload(noise_, nv.data(), n_ * 8);
        nvec = noise_;
    }
    dev_->zero(ymu_, npad_ * es_);
    dev_->upload(ymu_, ymu_host, n_ * es_);
    dev_->info(true);
    for (int i = 0; i < nown_; ++i) {  // cov! + nugget, own block-rows only (lower tiles; identity padding past n)
        int64_t ld, width;
        char* blk = block_ptr(i, &ld, &width);
        dev_->assemble(x_, n_, d_, (int64_t)own_[i] * WD_, WD_, nugget, nvec, blk, ld, width);
    }
    {
        int64_t ldc;
        char* cr = carried_ptr(&ldc);
        dev_->copy2d(cr, ldc * es_, ymu_, npad_ * es_, npad_ * es_, 1);
    }
    // whole compute units for the chain and the exchange when collectives must find room beside the update (G > 1) or the
    // update is short enough for the chain to be exposed; free slots beside a full-width update otherwise (common.h).  Chosen
    // per step below: a factorisation starts with free slots and ends on whole CUs.
    bool whole_now = G_ > 1 || npad_ - 2 * WD_ < kWholeCusBelow;
    dev_->whole_cus(whole_now);
    DevEvent e0 = dev_->record();
    for (DevStream s : {DS_SIDE, DS_COMM, DS_UPD}) {
        dev_->use(s);
        dev_->wait(e0);
    }
    DevEvent after = e0;
    if (rank_ == 0) {  // the first diagonal block has nothing to hide behind
        int64_t ld, width;
        char* blk = block_ptr(0, &ld, &width);
        dev_->super_factor(blk, ld, WD_, linv_, invd_, LW_, 0);
        after = dev_->record();
    }
    bcast_lw(0, after);
    solve_and_gather(0, false);
    for (int64_t k = 0; k + 1 < nblk_; ++k) {
        if (comm_rc_) break;  // a collective failed (or the group was aborted): nothing after it can be right
        const int64_t k0 = k * WD_, k1 = k0 + WD_;
        const int nle = n_le(rank_, k);
        const bool whole = G_ > 1 || npad_ - (k + 2) * WD_ < kWholeCusBelow;
        if (whole != whole_now) {  // the two modes use different streams: join the step's streams, switch, fan out again
            join_on_main();
            DevEvent ej = dev_->record();
            dev_->whole_cus(whole);
            whole_now = whole;
            for (DevStream st : {DS_UPD, DS_SIDE, DS_COMM}) {
                dev_->use(st);
                dev_->wait(ej);
            }
        }
        dev_->use(DS_UPD);
        // U1: block column k+1 of every own row below block k
        const bool mine_next = (k + 1) % G_ == rank_;
        // The next diagonal block FIRST, and from the owner's OWN solved rows (X X' with both operands in the factor: the rows of block
        // k+1 in panel k are the owner's, solved and copied back by this stream): it does not wait for the panel exchange, so the chain
        // starts while the gather is still on its way — the exchange is off the chain's critical path altogether.
        // (one rank: only the next diagonal block goes first — the chain needs nothing else — and the rest of block column k+1
        //  is part of the one big update below, as in chol.h; with an exchange to hide, the whole column goes first so that the
        //  next panel can be solved and sent early)
        DevShape rect, lower;
        lower.mode = 1;
        DevEvent ev_diag = nullptr;  // the next diagonal block is current: all the chain waits for (the rest of U1 runs beside it)
        const std::vector<Piece> below = pieces(nle, true);
        dev_->phase(GPMI_PROF_STEP_U1, true);
        if (mine_next && !below.empty() && below[0].nb > 0) {
            const Piece& pc = below[0];
            dev_->gemm(pc.p + k1 * es_, pc.ld, pc.p + k0 * es_, pc.ld, pc.p + k0 * es_, pc.ld, WD_, WD_, WD_, lower, 0);
            ev_diag = dev_->record();
        }
        dev_->wait(ev_p_);
        bool first_piece = true;
        for (const Piece& pc : below) {
            int64_t M = (int64_t)pc.nb * WD_ + (pc.carried ? 1 : 0);
            char* rows = pc.p;
            if (first_piece && mine_next && pc.nb > 0) {  // the diagonal block went ahead
                rows += WD_ * pc.ld * es_;
                M -= WD_;
            }
            first_piece = false;
            if (G_ > 1 && M > 0) dev_->gemm(rows + k1 * es_, pc.ld, rows + k0 * es_, pc.ld, panel_rows(k, k + 1), ldP_, M, WD_, WD_, rect, 0);
        }
        dev_->phase(GPMI_PROF_STEP_U1, false);
        DevEvent ev_u1 = dev_->record();
        DevEvent chain = ev_u1;
        if (mine_next) {
            const int li = (int)((k + 1) / G_);
            int64_t ld, width;
            char* blk = block_ptr(li, &ld, &width) + k1 * es_;
            dev_->use(DS_SIDE);
            dev_->wait(ev_diag ? ev_diag : ev_u1);
            dev_->phase(GPMI_PROF_STEP_CHAIN, true);
            dev_->super_factor(blk, ld, WD_, linv_ + (int64_t)li * WD_ * 64 * es_, invd_ + (int64_t)li * WD_ * es_, LW_ + (k + 1) * WD_ * WD_ * es_,
                               k1);
            dev_->phase(GPMI_PROF_STEP_CHAIN, false);
            chain = dev_->record();
        }
        bcast_lw(k + 1, chain);
        // U2a: enough block columns to cover the chain and the broadcast, then the next panel, then the rest under the exchange
        // (one rank: nothing to exchange — the whole update hides the chain, the next panel is solved after it, as chol.h does)
        const int64_t rest = nblk_ - (k + 2);
        int64_t m = nblk_;
        if (G_ > 1 && u2a_div_ > 0) {
            m = std::min<int64_t>(nblk_, k + 2 + std::max<int64_t>(rest > 0 ? 1 : 0, (rest + u2a_div_ - 1) / u2a_div_));
        } else if (G_ > 1) {
            // block columns k+2 .. m-1: enough flops (every rank's share of column c: (nblk - c) / G blocks of rows x WD x WD x 2) to keep the
            // update kernel busy for u2a_cover_s_ at ~55 (fp64) / ~110 (fp32) TFLOP/s; the same m on every rank
            const double need = u2a_cover_s_ * (es_ == 8 ? 55e12 : 110e12);
            double got = 0.0;
            m = k + 2;
            while (m < nblk_) {
                got += 2.0 * ((double)(nblk_ - m) * (double)WD_ / (double)G_) * (double)WD_ * (double)WD_;
                ++m;
                if (got >= need) break;
            }
        }
        dev_->use(DS_UPD);
        dev_->phase(GPMI_PROF_STEP_U2A, true);
        if (G_ == 1)
            update_cols(k, k + 1, nblk_, k + 2);  // everything but the next diagonal block, in one launch per stripe
        else
            update_cols(k, k + 2, m, 0);
        dev_->phase(GPMI_PROF_STEP_U2A, false);
        solve_and_gather(k + 1, false);
        dev_->use(DS_UPD);
        if (G_ > 1) {
            dev_->phase(GPMI_PROF_STEP_U2B, true);
            update_cols(k, m, nblk_, 0);
            dev_->phase(GPMI_PROF_STEP_U2B, false);
        }
    }
    // join the streams on the main one
    join_on_main();
    // the FIRST failing pivot wins (ranks past it have been factoring garbage), as dpotrf reports it
    double piv = (double)dev_->info(false);
    if (piv <= 0) piv = 1e18;
    if (comm_ && G_ > 1) {
        // the ranks' local error state rides along (min over -1 / 0), so that EVERY rank takes the same branch before the
        // collectives of the logdet and the backward solve: one rank returning alone would leave the others blocked in them
        double pv[2] = {piv, (!dev_->err.empty() || comm_rc_) ? -1.0 : 0.0};
        comm_rc_ |= comm_->host_allreduce(pv, 2, 1);
        piv = pv[0];
        if (pv[1] < 0 && dev_->err.empty() && !comm_rc_) return fail(GPMI_EDEVICE, "gpmi_fit: another rank of the communicator reported a device error");
    }
    if ((rc = check_dev("gpmi_fit"))) return rc;
    if (piv < 1e17) {
        if (info_out) *info_out = (int64_t)piv;
        return fail(GPMI_ENOTPD, "matrix is not positive definite; Cholesky factorization failed");
    }
    // logdet = 2 sum log L_ii: local share + all-reduce
    double half = 0.0;
    for (int i = 0; i < nown_; ++i) {
        int64_t ld, width;
        char* blk = block_ptr(i, &ld, &width);
        half += dev_->logdiag_sum(blk, ld, WD_, (int64_t)own_[i] * WD_);
    }
    if (comm_ && G_ > 1) comm_rc_ |= comm_->host_allreduce(&half, 1, 0);
    logdet_ = 2.0 * half;
    // backward solve L' alpha = z: rank 0 starts from z (every rank carried y - mu), the others from 0
    {
        int64_t ldc;
        char* cr = carried_ptr(&ldc);
        if (rank_ == 0)
            dev_->copy2d(v_, npad_ * es_, cr, ldc * es_, npad_ * es_, 1);
        else
            dev_->zero(v_, npad_ * es_);
        backward_solve(v_, alpha_);
    }
    const double dot = dev_->dot(ymu_, alpha_, n_);
    if (alpha_out) dev_->download(alpha_out, alpha_, n_ * es_);
    dev_->sync();
    if ((rc = check_dev("gpmi_fit"))) return rc;
    const double mll = -(dot + logdet_ + LOG2PI * (double)n_) / 2.0;  // GPE.jl:210
    if (mll_out) *mll_out = mll;
    fitted_ = true;
    return GPMI_OK;
}

// L' out = v, block-rows in reverse.  v = this rank's share of the right-hand side minus sum_{solved blocks} L_b' out_b: the owner
// of block c needs the TOTAL of its WD entries (an all-reduce of WD numbers), solves through the block's explicit inverse, and
// folds L_c' out_c into its own v; one all-reduce at the end replicates out (every block was written by exactly one rank).
void BlockedGP::backward_solve(char* v, char* out) {
    dev_->zero(out, npad_ * es_);
    for (int64_t c = nblk_ - 1; c >= 0; --c) {
        if (comm_rc_) break;
        const int owner = (int)(c % G_);
        char* vc = v + c * WD_ * es_;
        if (G_ > 1) {
            dev_->copy2d(seg_, WD_ * es_, vc, WD_ * es_, WD_ * es_, 1);
            comm_rc_ |= comm_->all_reduce_sum(seg_, WD_, es_, dev_->native_stream());
            if (rank_ == owner) dev_->copy2d(vc, WD_ * es_, seg_, WD_ * es_, WD_ * es_, 1);
        }
        if (rank_ == owner) {
            const int li = (int)(c / G_);
            int64_t ld, width;
            char* blk = block_ptr(li, &ld, &width);
            dev_->bsolve_block(blk, ld, c * WD_, WD_, linv_ + (int64_t)li * WD_ * 64 * es_, LW_ + c * WD_ * WD_ * es_, v, out);
        }
    }
    if (G_ > 1) comm_rc_ |= comm_->all_reduce_sum(out, npad_, es_, dev_->native_stream());
}

int BlockedGP::factor_diag(void* out_host) {
    if (!fitted_) return fail(GPMI_EARG, "gpmi_factor_diag: no valid factorisation (call gpmi_fit first)");
    dev_->begin_call();
    dev_->use(DS_MAIN);
    // gather 1 / L_ii-free: the diagonal itself, one strided copy per own block into v (scratch), all-reduced
    dev_->zero(v_, npad_ * es_);
    for (int i = 0; i < nown_; ++i) {
        int64_t ld, width;
        char* blk = block_ptr(i, &ld, &width);
        const int64_t c0 = (int64_t)own_[i] * WD_;
        dev_->copy2d(v_ + c0 * es_, es_, blk + c0 * es_, (ld + 1) * es_, es_, WD_);
    }
    if (comm_ && G_ > 1) comm_rc_ |= comm_->all_reduce_sum(v_, npad_, es_, dev_->native_stream());
    dev_->download(out_host, v_, n_ * es_);
    dev_->sync();
    return check_dev("gpmi_factor_diag");
}

// Right-looking whitening of P rows held column-split (Rloc_: P x own column blocks): the owner of block k forms
// V_k = R_k LW_k' and broadcasts it (P x WD), visit(k) runs on every rank with V_k in Vk_, then every rank updates ITS column
// blocks c > k with its own rows of the factor, R_c -= V_k L_ck'.
template <typename F>
void BlockedGP::whiten_blocks(int64_t P, int64_t ldR, F visit) {
    const int64_t Ppad = (P + 127) / 128 * 128;
    DevShape rect;
    for (int64_t k = 0; k < nblk_; ++k) {
        if (comm_rc_) break;
        const int owner = (int)(k % G_);
        if (rank_ == owner) {
            const int li = (int)(k / G_);
            dev_->gemm(Vk_, ldP_, Rloc_ + (int64_t)li * WD_ * es_, ldR, LW_ + k * WD_ * WD_ * es_, WD_, P, WD_, WD_, rect, DG_OVERWRITE | DG_KEND_COL);
        }
        if (G_ > 1) comm_rc_ |= comm_->broadcast(Vk_, Ppad * ldP_ * es_, owner, dev_->native_stream());
        visit(k);
        const int first = n_le(rank_, k);
        for (const Piece& pc : pieces(first, false)) {  // own column blocks c > k:  R_c -= V_k L_ck'
            const int64_t Nc = (int64_t)pc.nb * WD_;
            if (Nc <= 0) continue;
            dev_->gemm(Rloc_ + (int64_t)pc.b0 * WD_ * es_, ldR, Vk_, ldP_, pc.p + k * WD_ * es_, pc.ld, P, Nc, WD_, rect, 0);
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------------
// predict_f (GP.jl:64-84): the test points' cross-covariance is split by COLUMNS like the factor's rows: a rank holds
// R[:, own blocks] (P x nown WD).  Right-looking whitening: the owner of block k forms V_k = R_k LW_k' and broadcasts it
// (P x WD: N P elements per predict in total — nobody re-gathers the factor); every rank then updates ITS column blocks
// c > k with its own rows of the factor, R_c -= V_k L_ck'.  sigma^2 and the full covariance accumulate from the V_k, which
// every rank sees; mu = m + K*' alpha is a partial product over the own columns + one all-reduce of P numbers.
// ------------------------------------------------------------------------------------------------------------------------
int BlockedGP::predict(const gpmi_kernel* kern, int64_t P, const void* xpred_host, const void* mean_host, int full_cov, void* mu_out,
                       void* var_out) {
    if (!fitted_) return fail(GPMI_EARG, "gpmi_predict: no valid factorisation (call gpmi_fit first)");
    comm_rc_ = 0;
    dev_->err.clear();
    dev_->begin_call();
    dev_->use(DS_MAIN);
    int n_hyp = 0, rc;
    if ((rc = dev_->set_kernel(kern, d_, &kdiag_, &n_hyp)) != GPMI_OK) return fail(rc, dev_->err);
    const int64_t Ppad = (P + 127) / 128 * 128;
    const int64_t ldR = padded(std::max<int64_t>((int64_t)nown_ * WD_, WD_));
    const int64_t ldK = padded(Ppad);
    if ((rc = grow(&xp_, &xp_cap_, P * d_ * es_))) return rc;
    if ((rc = grow(&Rloc_, &Rloc_cap_, Ppad * ldR * es_))) return rc;
    if ((rc = grow(&Vk_, &Vk_cap_, Ppad * ldP_ * es_))) return rc;
    if ((rc = grow(&small_, &small_cap_, (2 * Ppad + (int64_t)maxown_ * WD_) * es_ + 2 * Ppad * 8))) return rc;
    if (full_cov && (rc = grow(&Kpp_, &Kpp_cap_, Ppad * ldK * es_))) return rc;
    char* mean_d = small_;
    char* mu_d = small_ + Ppad * es_;
    char* aloc = small_ + 2 * Ppad * es_;
    double* s2acc = (double*)(aloc + (int64_t)maxown_ * WD_ * es_);
    dev_->upload(xp_, xpred_host, P * d_ * es_);
    dev_->zero(small_, small_cap_);
    if (rank_ == 0) dev_->upload(mean_d, mean_host, P * es_);  // the prior mean enters once
    dev_->zero(Rloc_, Ppad * ldR * es_);
    DevShape rect;
    // K*' restricted to the own column blocks (GP.jl:44), and alpha restricted the same way
    for (int i = 0; i < nown_; ++i) {
        const int64_t c0 = (int64_t)own_[i] * WD_;
        const int64_t nb = std::max<int64_t>(0, std::min<int64_t>(WD_, n_ - c0));
        if (nb > 0) dev_->cov_rows(xp_, P, x_ + c0 * d_ * es_, nb, d_, Rloc_ + (int64_t)i * WD_ * es_, ldR, WD_);
        dev_->copy2d(aloc + (int64_t)i * WD_ * es_, WD_ * es_, alpha_ + c0 * es_, WD_ * es_, WD_ * es_, 1);
    }
    dev_->row_gemv(Rloc_, ldR, P, (int64_t)nown_ * WD_, aloc, mean_d, mu_d);  // mu = mx + Kfx' alpha (GP.jl:26), this rank's share
    if (G_ > 1) comm_rc_ |= comm_->all_reduce_sum(mu_d, P, es_, dev_->native_stream());
    if (full_cov) dev_->cov_rows(xp_, P, xp_, P, d_, Kpp_, ldK, Ppad);
    whiten_blocks(P, ldR, [&](int64_t) {
        if (!full_cov)
            dev_->row_sumsq_acc(Vk_, ldP_, P, WD_, s2acc);
        else
            dev_->gemm(Kpp_, ldK, Vk_, ldP_, Vk_, ldP_, P, P, WD_, rect, 0);  // Kpred - Lck'Lck (GP.jl:45,51-54)
    });
    dev_->download(mu_out, mu_d, P * es_);
    if (!full_cov) {
        std::vector<double> s2((size_t)P);
        dev_->download(s2.data(), s2acc, P * 8);
        dev_->sync();
        if ((rc = check_dev("gpmi_predict"))) return rc;
        for (int64_t p = 0; p < P; ++p) {  // max(k** - |V|^2, 0): the clamp of GP.jl:75
            const double v = std::max(kdiag_ - s2[(size_t)p], 0.0);
            if (es_ == 8)
                ((double*)var_out)[p] = v;
            else
                ((float*)var_out)[p] = (float)v;
        }
        return GPMI_OK;
    }
    std::vector<char> tmp((size_t)(P * ldK * es_));
    dev_->download(tmp.data(), Kpp_, P * ldK * es_);
    dev_->sync();
    if ((rc = check_dev("gpmi_predict"))) return rc;
    for (int64_t p = 0; p < P; ++p) memcpy((char*)var_out + p * P * es_, tmp.data() + p * ldK * es_, (size_t)(P * es_));
    return GPMI_OK;
}

// V_own = I_own L^-T into G1_ (own rows x npad): row block i of the identity is zero left of column own[i] WD, so block column k
// only concerns the own blocks with global index <= k.  The factor's panels are re-gathered from the stored factor, as in the
// factorisation (UPD packs, COMM gathers).  Used by update_dmll! (K^-1 = V'V block by block) and by inv_diag (row norms of V).
int BlockedGP::whiten_identity_own() {
    int rc;
    const int64_t ldG = padded(npad_);
    const int64_t own_rows = (int64_t)std::max(nown_, 1) * WD_;
    if ((rc = grow(&G1_, &G1_cap_, own_rows * ldG * es_))) return rc;
    if ((rc = grow(&Wt_, &Wt_cap_, own_rows * ldP_ * es_))) return rc;
    for (int i = 0; i < nown_; ++i) dev_->set_identity_rows(G1_ + (int64_t)i * WD_ * ldG * es_, ldG, WD_, (int64_t)own_[i] * WD_);
    dev_->whole_cus(G_ > 1);
    DevEvent e0 = dev_->record();
    dev_->use(DS_UPD);
    dev_->wait(e0);
    dev_->use(DS_COMM);
    dev_->wait(e0);
    DevShape rect;
    char* S2 = Wt_;  // out-of-place image of the solved columns (own_rows x ldP)
    for (int64_t k = 0; k < nblk_; ++k) {
        if (comm_rc_) break;
        const int nrows_blk = n_le(rank_, k);  // own blocks <= k
        const int64_t M = (int64_t)nrows_blk * WD_;
        if (k + 1 < nblk_) solve_and_gather(k, true);  // P_k from the stored factor
        dev_->use(DS_UPD);
        if (M > 0) {
            dev_->gemm(S2, ldP_, G1_ + k * WD_ * es_, ldG, LW_ + k * WD_ * WD_ * es_, WD_, M, WD_, WD_, rect, DG_OVERWRITE | DG_KEND_COL);
            dev_->copy2d(G1_ + k * WD_ * es_, ldG * es_, S2, ldP_ * es_, WD_ * es_, M);
        }
        if (k + 1 < nblk_) {
            dev_->wait(ev_p_);
            if (M > 0)
                dev_->gemm(G1_ + (k + 1) * WD_ * es_, ldG, G1_ + k * WD_ * es_, ldG, panel_rows(k, k + 1), ldP_, M, npad_ - (k + 1) * WD_, WD_, rect, 0);
        }
    }
    join_on_main();
    return GPMI_OK;
}

// ------------------------------------------------------------------------------------------------------------------------
// The AbstractPDMat surface on a blocked handle (VERDICT r3 missing 2).  PDMats `\` (GPE.jl:208: update_mll!(kern = false,
// noise = false)), whiten! (GP.jl:27), unwhiten / rand (GP.jl:120-146), inv(cK) diag (crossvalidation.jl:8-13), cholfactors
// (GP.jl:89).  b is n x nrhs column-major == nrhs rows of n: the rows are whitened exactly like predict_f's cross-covariance
// rows (column-split over the ranks, V_k broadcast), the whitened blocks are collected on every rank, and the backward half
// runs the fit's distributed back-substitution once per right-hand side.
// ------------------------------------------------------------------------------------------------------------------------
int BlockedGP::solve(int64_t nrhs, void* b_host, bool backward) {
    if (!fitted_) return fail(GPMI_EARG, "gpmi_solve / gpmi_whiten: no valid factorisation (call gpmi_fit first)");
    if (nrhs <= 0 || !b_host) return fail(GPMI_EARG, "gpmi_solve / gpmi_whiten: bad argument");
    comm_rc_ = 0;
    dev_->err.clear();
    dev_->begin_call();
    dev_->use(DS_MAIN);
    int rc;
    const int64_t P = nrhs, Ppad = (P + 127) / 128 * 128;
    const int64_t ldR = padded(std::max<int64_t>((int64_t)nown_ * WD_, WD_));
    const int64_t ldB = padded(npad_);
    if ((rc = grow(&Rloc_, &Rloc_cap_, Ppad * ldR * es_))) return rc;
    if ((rc = grow(&Vk_, &Vk_cap_, Ppad * ldP_ * es_))) return rc;
    if ((rc = grow(&Bfull_, &Bfull_cap_, Ppad * ldB * es_))) return rc;
    if ((rc = grow(&Bc_, &Bc_cap_, P * n_ * es_))) return rc;
    dev_->zero(Bfull_, Ppad * ldB * es_);
    dev_->zero(Rloc_, Ppad * ldR * es_);
    dev_->upload(Bc_, b_host, P * n_ * es_);
    dev_->copy2d(Bfull_, ldB * es_, Bc_, n_ * es_, n_ * es_, P);
    for (int i = 0; i < nown_; ++i)  // the rows' own column blocks
        dev_->copy2d(Rloc_ + (int64_t)i * WD_ * es_, ldR * es_, Bfull_ + (int64_t)own_[i] * WD_ * es_, ldB * es_, WD_ * es_, P);
    whiten_blocks(P, ldR, [&](int64_t k) { dev_->copy2d(Bfull_ + k * WD_ * es_, ldB * es_, Vk_, ldP_ * es_, WD_ * es_, P); });
    if (backward) {
        for (int64_t r = 0; r < P; ++r) {
            if (comm_rc_) break;
            char* row = Bfull_ + r * ldB * es_;
            if (rank_ == 0)
                dev_->copy2d(v_, npad_ * es_, row, npad_ * es_, npad_ * es_, 1);
            else
                dev_->zero(v_, npad_ * es_);
            backward_solve(v_, row);  // (seg_ and v_ are scratch; alpha_ — the fit's — is not touched)
        }
    }
    dev_->copy2d(Bc_, n_ * es_, Bfull_, ldB * es_, n_ * es_, P);
    dev_->download(b_host, Bc_, P * n_ * es_);
    dev_->sync();
    return check_dev(backward ? "gpmi_solve" : "gpmi_whiten");
}

int BlockedGP::update_alpha(const void* ymu_host, double* mll_out, void* alpha_out) {
    if (!fitted_) return fail(GPMI_EARG, "gpmi_update_alpha: no valid factorisation (call gpmi_fit first)");
    if (!ymu_host) return fail(GPMI_EARG, "gpmi_update_alpha: bad argument");
    std::vector<char> b((size_t)(n_ * es_));
    std::memcpy(b.data(), ymu_host, b.size());
    const int rc = solve(1, b.data(), true);  // every rank receives the same alpha
    if (rc) return rc;
    dev_->begin_call();
    dev_->use(DS_MAIN);
    dev_->zero(ymu_, npad_ * es_);
    dev_->upload(ymu_, ymu_host, n_ * es_);
    dev_->zero(alpha_, npad_ * es_);
    dev_->upload(alpha_, b.data(), n_ * es_);
    const double dot = dev_->dot(ymu_, alpha_, n_);
    dev_->sync();
    if (alpha_out) std::memcpy(alpha_out, b.data(), b.size());
    if (mll_out) *mll_out = -(dot + logdet_ + LOG2PI * (double)n_) / 2.0;  // GPE.jl:210
    return check_dev("gpmi_update_alpha");
}

int BlockedGP::inv_diag(void* out_host) {
    if (!fitted_) return fail(GPMI_EARG, "gpmi_inv_diag: no valid factorisation (call gpmi_fit first)");
    comm_rc_ = 0;
    dev_->err.clear();
    dev_->begin_call();
    dev_->use(DS_MAIN);
    int rc;
    if ((rc = whiten_identity_own())) return rc;  // rows of V = L^-T; (K^-1)_ii = |V_i|^2
    const int64_t ldG = padded(npad_);
    const int64_t own_rows = (int64_t)std::max(nown_, 1) * WD_;
    if ((rc = grow((char**)&dfull_, &dfull_cap_, (npad_ + own_rows) * 8))) return rc;
    double* acc = dfull_ + npad_;
    dev_->zero(dfull_, (npad_ + own_rows) * 8);
    if (nown_ > 0) dev_->row_sumsq_acc(G1_, ldG, (int64_t)nown_ * WD_, npad_, acc);
    for (int i = 0; i < nown_; ++i) dev_->copy2d(dfull_ + (int64_t)own_[i] * WD_, WD_ * 8, acc + (int64_t)i * WD_, WD_ * 8, WD_ * 8, 1);
    if (G_ > 1) comm_rc_ |= comm_->all_reduce_sum(dfull_, npad_, 8, dev_->native_stream());
    std::vector<double> h((size_t)n_);
    dev_->download(h.data(), dfull_, n_ * 8);
    dev_->sync();
    if ((rc = check_dev("gpmi_inv_diag"))) return rc;
    for (int64_t i = 0; i < n_; ++i) {
        if (es_ == 8)
            ((double*)out_host)[i] = h[(size_t)i];
        else
            ((float*)out_host)[i] = (float)h[(size_t)i];
    }
    return GPMI_OK;
}

int BlockedGP::factor_to_host(void* U_out) {
    if (!fitted_) return fail(GPMI_EARG, "gpmi_factor_to_host: no valid factorisation (call gpmi_fit first)");
    comm_rc_ = 0;
    dev_->err.clear();
    dev_->begin_call();
    dev_->use(DS_MAIN);
    int rc;
    const int64_t ldG = padded(npad_);
    if ((rc = grow(&Vb_, &Vb_cap_, WD_ * ldG * es_))) return rc;
    std::vector<char> tmp((size_t)(WD_ * ldG * es_));
    char* o = (char*)U_out;
    for (int64_t b = 0; b < nblk_; ++b) {
        if (comm_rc_) break;
        const int owner = (int)(b % G_);
        const int64_t r0 = b * WD_, nr = std::max<int64_t>(0, std::min<int64_t>(WD_, n_ - r0));
        if (nr == 0) break;
        const int64_t w = r0 + WD_;  // the block-row's columns up to its own diagonal block
        if (rank_ == owner) {
            int64_t ld, width;
            char* blk = block_ptr((int)(b / G_), &ld, &width);
            dev_->copy2d(Vb_, ldG * es_, blk, ld * es_, w * es_, WD_);
        }
        if (G_ > 1) comm_rc_ |= comm_->broadcast(Vb_, WD_ * ldG * es_, owner, dev_->native_stream());
        dev_->download(tmp.data(), Vb_, WD_ * ldG * es_);
        // row-major lower L == column-major upper U (GPE.jl:60): row i holds its first i + 1 entries, zeros after them
        for (int64_t i = 0; i < nr; ++i) {
            const int64_t gi = r0 + i;
            memcpy(o + (size_t)gi * n_ * es_, tmp.data() + (size_t)i * ldG * es_, (size_t)(gi + 1) * es_);
            if (gi + 1 < n_) memset(o + ((size_t)gi * n_ + gi + 1) * es_, 0, (size_t)(n_ - 1 - gi) * es_);
        }
    }
    dev_->sync();
    return check_dev("gpmi_factor_to_host");
}

// ------------------------------------------------------------------------------------------------------------------------
// update_dmll! (GPE.jl:298-324).  K^-1 = L^-T L^-1 = V V' with V = L^-T, whose ROWS are the whitened identity rows.  A rank
// whitens the identity rows of the blocks it owns (phase 1: the factor's panels are re-gathered, as in the factorisation),
// then the block-rows of V are broadcast one by one (phase 2) and every rank forms the blocks K^-1[own rows i >= b, block b]
// it needs, turns them into W = alpha alpha' - K^-1 and reduces <W, dK/dtheta> through the fused trace kernel — K^-1 is never
// resident: the extra memory is ONE own-rows x N matrix (N^2 / G), not the two N x N of gpmi_grad.
// ------------------------------------------------------------------------------------------------------------------------
int BlockedGP::grad(const gpmi_kernel* kern, const double* log_noise, int64_t n_noise, double* dkern_out, int n_kern, double* dnoise_out) {
    if (!fitted_) return fail(GPMI_EARG, "gpmi_grad: no valid factorisation (call gpmi_fit first)");
    if (n_noise != 1 && dnoise_out) return fail(GPMI_EARG, "gpmi_grad: the noise gradient is defined for scalar logNoise only (GPE.jl:313)");
    comm_rc_ = 0;
    dev_->err.clear();
    dev_->begin_call();
    dev_->use(DS_MAIN);
    int n_hyp = 0, rc;
    if ((rc = dev_->set_kernel(kern, d_, &kdiag_, &n_hyp)) != GPMI_OK) return fail(rc, dev_->err);
    if (n_hyp != n_kern) return fail(GPMI_EARG, "gpmi_grad: dkern_out length differs from the kernel's number of parameters");
    if (n_hyp > GPMI_GRAD_MAX_PARAMS)
        return fail(GPMI_EARG, "gpmi_grad: more than GPMI_GRAD_MAX_PARAMS (5000) kernel hyper-parameters: the trace kernel's per-wave table does not fit the LDS");
    const int64_t ldG = padded(npad_);
    const int64_t own_rows = (int64_t)std::max(nown_, 1) * WD_;
    if ((rc = grow(&Vb_, &Vb_cap_, WD_ * ldG * es_))) return rc;
    if ((rc = grow((char**)&dacc_, &dacc_cap_, (int64_t)(n_hyp + 2) * 8))) return rc;
    if (!xloc_) {  // the own rows' inputs and alpha, contiguous in local order
        xloc_ = (char*)grab(own_rows * d_ * es_, true);
        aloc_ = (char*)grab(own_rows * es_, true);
        if (!xloc_ || !aloc_) return fail(GPMI_EDEVICE, "blocked GP: out of device memory");
        for (int i = 0; i < nown_; ++i) {
            const int64_t c0 = (int64_t)own_[i] * WD_, nb = std::max<int64_t>(0, std::min<int64_t>(WD_, n_ - c0));
            if (nb > 0) dev_->copy2d(xloc_ + (int64_t)i * WD_ * d_ * es_, nb * d_ * es_, x_ + c0 * d_ * es_, nb * d_ * es_, nb * d_ * es_, 1);
        }
    }
    for (int i = 0; i < nown_; ++i)
        dev_->copy2d(aloc_ + (int64_t)i * WD_ * es_, WD_ * es_, alpha_ + (int64_t)own_[i] * WD_ * es_, WD_ * es_, WD_ * es_, 1);
    dev_->zero(dacc_, (int64_t)(n_hyp + 2) * 8);
    if ((rc = whiten_identity_own())) return rc;  // phase 1: G1_ = the own blocks' rows of V = L^-T
    DevShape rect;
    // ---- phase 2: block-rows of V broadcast in turn; K^-1[own rows of blocks >= b, block b] = V_i V_b' (K from the later of
    //      the two diagonals), W = w (alpha alpha' - K^-1) with w = 1 below the diagonal block and 1/2 on it, trace kernel.
    for (int64_t b = 0; b < nblk_; ++b) {
        if (comm_rc_) break;
        const int owner = (int)(b % G_);
        const int64_t b0 = b * WD_, nbc = std::max<int64_t>(0, std::min<int64_t>(WD_, n_ - b0));
        const char* Vb = nullptr;
        if (G_ == 1) {
            Vb = G1_ + b * WD_ * ldG * es_;
        } else {
            if (rank_ == owner) dev_->copy2d(Vb_, ldG * es_, G1_ + (b / G_) * WD_ * ldG * es_, ldG * es_, ldG * es_, WD_);
            comm_rc_ |= comm_->broadcast(Vb_, WD_ * ldG * es_, owner, dev_->native_stream());
            Vb = Vb_;
        }
        if (nbc == 0) continue;
        // own blocks with global index >= b, in up to four chunks: one product per chunk whose K loop starts at the chunk's
        // first diagonal (rows further down are zero there: wasted flops bounded by the chunk's extent)
        const int first = n_le(rank_, b - 1);
        const int cnt = nown_ - first;
        const int nch = std::min(4, cnt);
        for (int c = 0; c < nch; ++c) {
            const int i0 = first + cnt * c / nch, i1 = first + cnt * (c + 1) / nch;
            if (i1 <= i0) continue;
            const int64_t M = (int64_t)(i1 - i0) * WD_;
            const int64_t ks = (int64_t)own_[i0] * WD_;  // >= b0
            dev_->gemm(Wt_, ldP_, G1_ + ((int64_t)i0 * WD_ * ldG + ks) * es_, ldG, Vb + ks * es_, ldG, M, WD_, npad_ - ks, rect, DG_OVERWRITE);
            const bool diag = own_[i0] == b;  // the chunk's first block is the diagonal block
            int ia = i0;
            if (diag) {
                dev_->qblock(Wt_, ldP_, nbc, nbc, aloc_ + (int64_t)i0 * WD_ * es_, alpha_ + b0 * es_, 0.5, false, true, nbc, dacc_ + n_hyp);
                dev_->dmll_rect_acc(xloc_ + (int64_t)i0 * WD_ * d_ * es_, nbc, x_ + b0 * d_ * es_, nbc, d_, Wt_, ldP_, n_hyp, dacc_);
                ia = i0 + 1;
            }
            // the chunk's rows below the diagonal block in one pass: real observations only (they are a prefix: only the last
            // global block carries identity padding, which has no kernel entries)
            int64_t nr = 0;
            for (int i = ia; i < i1; ++i) nr += std::max<int64_t>(0, std::min<int64_t>(WD_, n_ - (int64_t)own_[i] * WD_));
            if (nr > 0) {
                char* Wi = Wt_ + (int64_t)(ia - i0) * WD_ * ldP_ * es_;
                dev_->qblock(Wi, ldP_, nr, nbc, aloc_ + (int64_t)ia * WD_ * es_, alpha_ + b0 * es_, 1.0, false, false, 0, nullptr);
                dev_->dmll_rect_acc(xloc_ + (int64_t)ia * WD_ * d_ * es_, nr, x_ + b0 * d_ * es_, nbc, d_, Wi, ldP_, n_hyp, dacc_);
            }
        }
    }
    std::vector<double> h((size_t)n_hyp + 2);
    dev_->download(h.data(), dacc_, (int64_t)(n_hyp + 2) * 8);
    dev_->sync();
    if ((rc = check_dev("gpmi_grad"))) return rc;
    if (comm_ && G_ > 1)  // in pieces: a communicator's host reduction carries a bounded number of doubles per call
        for (int p0 = 0; p0 < n_hyp + 1; p0 += 64) comm_rc_ |= comm_->host_allreduce(h.data() + p0, std::min(64, n_hyp + 1 - p0), 0);
    if ((rc = check_dev("gpmi_grad"))) return rc;
    for (int p = 0; p < n_hyp; ++p) dkern_out[p] = h[(size_t)p];
    if (dnoise_out) *dnoise_out = exp(2.0 * log_noise[0]) * h[(size_t)n_hyp];  // GPE.jl:273-275
    return GPMI_OK;
}

}  // namespace gpmi
