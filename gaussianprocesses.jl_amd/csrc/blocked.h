// blocked.h — the BLOCKED exact GP: the factor of K + noise as block-rows of WD = 256 * 2^s rows, dealt round-robin over
// the ranks of a communicator (row-block sharding) and, per rank, kept in stripes that stop at their own
// diagonal (packed storage).  One rank + packed stripes = a single device past the N x N ceiling; G ranks =
// one process per GPU with the panel exchange over RCCL.  Host-side driver only: every flop goes through dev.h.
//
// Reference semantics: update_cK! / update_mll! (src/GPE.jl:169-212), make_posdef! (src/GP.jl:101-112), predict_f /
// predictMVN! (src/GP.jl:25-84), update_dmll! (src/GPE.jl:298-324, dmll_kern! :219-241, dmll_noise :273-275).
#pragma once
#include <memory>
#include <vector>

#include "dev.h"

namespace gpmi {

struct BlockedOpts {
    int64_t block = 0;      // rows per distributed block (0: 1024 from 16 384 points — 2048 from 131 072 on one rank —, 512 from 4096, 256 below)
    int stripe_blocks = 0;  // local blocks per storage stripe (0: one stripe = the plain rows x npad matrix)
};

class BlockedGP {
  public:
    BlockedGP(Dev* dev, Comm* comm, int d, int64_t n, BlockedOpts o);
    ~BlockedGP();
    int init(const void* x_rowmajor_host);  // allocations + upload of x (n x d row-major == Julia's d x n)
    int fit(const gpmi_kernel* k, const double* log_noise, int64_t n_noise, const void* ymu_host, double* mll_out, void* alpha_out,
            int64_t* info_out);
    int predict(const gpmi_kernel* k, int64_t P, const void* xpred_host, const void* mean_host, int full_cov, void* mu_out, void* var_out);
    int grad(const gpmi_kernel* k, const double* log_noise, int64_t n_noise, double* dkern_out, int n_kern, double* dnoise_out);
    int factor_diag(void* out_host);  // diag(U), n elements (every rank returns the full vector)
    // AbstractPDMat surface (PDMats `\`, whiten!; GPE.jl:208, GP.jl:27,136, GPE.jl:162).  b: n x nrhs column-major on the host, overwritten
    // with L^-1 b (backward = false) or (K + noise)^-1 b (true); every rank passes the same b and receives the same result
    int solve(int64_t nrhs, void* b_inout_host, bool backward);
    // update_mll!(noise = false, kern = false), GPE.jl:203-211: alpha = cK \ (y - mu) through the kept factor; replaces the device copies of
    // alpha and y - mu (predict / grad read them) and returns the mll from the stored logdet
    int update_alpha(const void* ymu_host, double* mll_out, void* alpha_out);
    int inv_diag(void* out_host);        // diag((K + noise)^-1), n elements (crossvalidation.jl:8-13); needs the gradient's own-rows x N scratch
    int factor_to_host(void* U_out_host);  // n x n column-major upper factor, zeros below (GPE.jl:60); every rank receives all of it
    bool fitted() const { return fitted_; }
    double logdet() const { return logdet_; }
    const std::string& error() const { return err_; }
    int64_t nobs() const { return n_; }
    int64_t block_rows() const { return WD_; }
    int64_t stored_bytes() const { return stored_bytes_; }  // the factor's device footprint on this rank
    int nstripes() const { return (int)stripes_.size(); }

  private:
    struct Stripe {
        int i0, i1;        // local blocks [i0, i1)
        char* p;           // rows x ld elements
        int64_t ld, width, rows;
    };
    struct Piece {
        char* p;           // first row of the piece
        int64_t ld, width;
        int b0, nb;        // first local block, number of blocks
        bool carried;      // the piece ends with the carried (y - mu) row
    };
    Dev* dev_;
    Comm* comm_;
    int rank_, G_, d_;
    int64_t n_, WD_, npad_, nblk_, ldP_;
    int es_, tpb_, per_;
    std::vector<int> own_;
    int nown_ = 0, maxown_ = 0;
    std::vector<Stripe> stripes_;
    std::vector<void*> allocs_;
    int64_t stored_bytes_ = 0;
    char *x_ = nullptr, *LW_ = nullptr, *linv_ = nullptr, *invd_ = nullptr, *alpha_ = nullptr, *v_ = nullptr, *ymu_ = nullptr, *seg_ = nullptr;
    char *S_[2] = {nullptr, nullptr}, *P_[2] = {nullptr, nullptr};
    double* noise_ = nullptr;
    // predict / gradient scratch, grown on demand
    char *xp_ = nullptr, *Rloc_ = nullptr, *Vk_ = nullptr, *small_ = nullptr, *Kpp_ = nullptr, *aloc_ = nullptr, *xloc_ = nullptr;
    int64_t xp_cap_ = 0, Rloc_cap_ = 0, Vk_cap_ = 0, small_cap_ = 0, Kpp_cap_ = 0;
    double* dacc_ = nullptr;
    int64_t dacc_cap_ = 0;
    char *G1_ = nullptr, *Vb_ = nullptr, *Wt_ = nullptr;
    int64_t G1_cap_ = 0, Vb_cap_ = 0, Wt_cap_ = 0;
    bool fitted_ = false;
    double logdet_ = 0.0, kdiag_ = 0.0;
    std::string err_;

    // layout helpers
    int64_t padded(int64_t ncols) const { return (ncols * es_) % 4096 == 0 ? ncols + 64 : ncols; }
    int n_le(int q, int64_t k) const { return k >= q ? (int)((k - q) / G_ + 1) : 0; }  // blocks of rank q with global index <= k
    int n_below(int q, int64_t k) const { return n_own_of(q) - n_le(q, k); }
    int n_own_of(int q) const { return q < nblk_ ? (int)((nblk_ - 1 - q) / G_ + 1) : 0; }
    const Stripe& stripe_of(int i) const;
    char* block_ptr(int i, int64_t* ld, int64_t* width) const;
    char* carried_ptr(int64_t* ld) const;
    std::vector<Piece> pieces(int first_block, bool carried) const;
    char* panel_rows(int64_t k, int64_t b) const;  // rows of global block b (> k) in the gathered panel k

    void* grab(int64_t bytes, bool zero);
    int grow(char** p, int64_t* cap, int64_t bytes);
    int fail(int rc, const std::string& msg) { err_ = msg; return rc; }
    int check_dev(const char* where);

    // factorisation pieces
    DevEvent ev_lw_ = nullptr, ev_p_ = nullptr;
    void bcast_lw(int64_t k, DevEvent after);
    void solve_and_gather(int64_t k, bool from_factor);  // from_factor: the rows are already solved (gradient: re-gather a stored panel)
    void update_cols(int64_t k, int64_t c_lo, int64_t c_hi, int64_t min_block);
    void join_on_main();
    void backward_solve(char* v, char* out);  // L' out = v over the block-rows in reverse (v: rank 0 holds the right-hand side, the others zeros; consumed)
    template <typename F>
    void whiten_blocks(int64_t P, int64_t ldR, F visit);  // Rloc_ (P x own columns) <- rows whitened block by block; visit(k) sees V_k (P x WD) in Vk_
    int whiten_identity_own();                // G1_ <- the own blocks' rows of L^-T (the whitened identity rows; the factor's panels re-gathered)
    char *Bfull_ = nullptr, *Bc_ = nullptr;
    int64_t Bfull_cap_ = 0, Bc_cap_ = 0;
    double* dfull_ = nullptr;
    int64_t dfull_cap_ = 0;
    int comm_rc_ = 0;
    // U2a = the part of a step's update that hides the chain and the inverse broadcast; the panel exchange hides under the rest (U2b), so U2a
    // should be just long enough.  Rounds 3-4: a fixed half of the remaining block columns (u2a_div_ = 2: 367 / 362 / 344 ms for 4 / 3 / 2 on
    // two CU partitions at N = 32 768 with the 3.5 ms multi-launch chain, profiles/r04_i_partitions.log) — but half of the COLUMNS of a lower
    // staircase is three quarters of its area, which left the exchange a quarter of the update to hide under.  Round 5: as many block columns
    // as cover the chain kernel's ~1.1 ms (x (WD / 1024)^2) plus the broadcast at the update kernel's rate, counted in flops.
    int u2a_div_ = 0;            // > 0: the fixed fraction again (GPMI_BLOCKED_U2A, tools builds)
    double u2a_cover_s_ = 0.0;   // seconds of update that U2a must hold (set in the constructor from WD)
};

}  // namespace gpmi
