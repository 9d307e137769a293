# GPMI355X.jl — the Julia-side binding a GaussianProcesses.jl maintainer adds to route the
# exact-GP hot path (update_mll! / predict_f) through libgpmi.so on an MI355X.
#
# STATUS: written against include/gpmi.h, NOT executed — the build container has no Julia
# (SURVEY.md header).  The same ABI is exercised end-to-end by the Python/ctypes host mirror
# (gaussianprocesses.jl_amd/gpmi355x) and the GPU parity tests, so this file only has to be a
# thin, reviewable marshaling layer.  Because it has never been parsed, every method below was
# audited by eye against the reference methods of the same name it competes with in dispatch
# (INTEGRATION.md "Dispatch audit": ambiguity, keyword and type-parameter mismatches); the test
# a maintainer runs on day one is julia/runtests.jl.
# It plugs in exactly where the reference's own alternative strategies do (SoR/DTC/FITC/FSA: src/sparse/subsetofregressors.jl:82-113,302-327):
#   * a CovarianceStrategy subtype selected through the 6-argument GPE constructor (src/GPE.jl:68-71)
#   * an AbstractPDMat subtype returned by alloc_cK (src/GP.jl:14-20)
#   * methods for update_cK!, \, logdet, whiten!, predictMVN, predict_f (src/GPE.jl:169-212, src/GP.jl:25-84)
module GPMI355X

using GaussianProcesses
using GaussianProcesses: GPE, GPBase, Kernel, Mean, EmptyData, CovarianceStrategy,
    SEIso, SEArd, Mat12Iso, Mat12Ard, Mat32Iso, Mat32Ard, Mat52Iso, Mat52Ard, RQIso, RQArd,
    Noise, Const, SumKernel, ProdKernel, Masked, FixedKernel, get_value, AbstractGradientPrecompute, SparseStrategy, SparsePDMat
import GaussianProcesses: alloc_cK, update_cK!, update_mll!, num_params, get_alpha_u, predictMVN, predict_f, mat, cholfactors,
    init_precompute, precompute!, dmll_kern!, dmll_noise, predict_LOO, KernelData
using Statistics: mean                 # GaussianProcesses extends Statistics.mean for mean(m::Mean, X)  (src/GaussianProcesses.jl:8)
using PDMats
import PDMats: dim, whiten!, whiten, unwhiten!
using LinearAlgebra
import LinearAlgebra: logdet, \, ldiv!, tr

const LOG2PI = log(2π)                 # StatsFuns.log2π inside the reference (src/GPE.jl:210)
const libgpmi = get(ENV, "LIBGPMI", "libgpmi.so")

# ---- return codes -> Julia exceptions (include/gpmi.h "Conventions") ----------------------
function check(ctx::Ptr{Cvoid}, rc::Cint, info::Integer=0)
    rc == 0 && return
    rc == 1 && throw(LinearAlgebra.PosDefException(info))          # src/GP.jl:110; caught by src/optimize.jl:56-58
    msg = unsafe_string(ccall((:gpmi_last_error, libgpmi), Cstring, (Ptr{Cvoid},), ctx))
    rc == 2 && throw(ArgumentError(msg))
    error("libgpmi: $msg")
end

# ENV["GPMI_DEVICE"] = "3"           one GPU (default 0)
# ENV["GPMI_DEVICES"] = "0,1,2,3"      an in-process DEVICE GROUP (gpmi_ctx_create with n_devices > 1): blocked models —
#                                      HIPCovariance(sharded=true) / (packed=true) — are row-block sharded over these GPUs by
#                                      worker threads inside libgpmi; this one Julia session drives them all, no MPI
const CTX = Ref{Ptr{Cvoid}}(C_NULL)
function context()
    if CTX[] == C_NULL
        h = Ref{Ptr{Cvoid}}(C_NULL)
        devs = haskey(ENV, "GPMI_DEVICES") ? Cint[parse(Cint, v) for v in split(ENV["GPMI_DEVICES"], ",")] :
                                             Cint[parse(Cint, get(ENV, "GPMI_DEVICE", "0"))]
        rc = ccall((:gpmi_ctx_create, libgpmi), Cint, (Cint, Ptr{Cint}, Ptr{Ptr{Cvoid}}), length(devs), devs, h)
        rc == 0 || error("libgpmi: no usable MI355X (gpmi_ctx_create rc=$rc); there is no CPU backend")
        CTX[] = h[]
    end
    CTX[]
end

# ---- kernel tree -> gpmi_kernel postfix descriptor -----------------------------------------
struct KernelDesc
    ops::Vector{Int32}; dims_off::Vector{Int32}; dims::Vector{Int32}; params::Vector{Float64}
end
KernelDesc() = KernelDesc(Int32[], Int32[0], Int32[], Float64[])
struct CKernel   # mirrors `struct gpmi_kernel`
    n_ops::Int32; ops::Ptr{Int32}; dims_off::Ptr{Int32}; dims::Ptr{Int32}; params::Ptr{Float64}; n_params::Int32
end
function leaf!(kd, code, active, stored)
    push!(kd.ops, code)
    active === nothing || append!(kd.dims, Int32.(active .- 1))      # 0-based on the C side
    push!(kd.dims_off, length(kd.dims))
    append!(kd.params, stored)
end
flatten!(kd, k::SEIso, a)    = leaf!(kd, 1, a, [k.ℓ2, k.σ2])
flatten!(kd, k::SEArd, a)    = leaf!(kd, 2, a, [k.iℓ2; k.σ2])
flatten!(kd, k::Mat12Iso, a) = leaf!(kd, 3, a, [k.ℓ, k.σ2])
flatten!(kd, k::Mat12Ard, a) = leaf!(kd, 4, a, [k.iℓ2; k.σ2])
flatten!(kd, k::Mat32Iso, a) = leaf!(kd, 5, a, [k.ℓ, k.σ2])
flatten!(kd, k::Mat32Ard, a) = leaf!(kd, 6, a, [k.iℓ2; k.σ2])
flatten!(kd, k::Mat52Iso, a) = leaf!(kd, 7, a, [k.ℓ, k.σ2])
flatten!(kd, k::Mat52Ard, a) = leaf!(kd, 8, a, [k.iℓ2; k.σ2])
flatten!(kd, k::RQIso, a)    = leaf!(kd, 9, a, [k.ℓ2, k.σ2, k.α])
flatten!(kd, k::RQArd, a)    = leaf!(kd, 10, a, [k.iℓ2; k.σ2; k.α])
flatten!(kd, k::Noise, a)    = leaf!(kd, 11, a, [k.σ2])
flatten!(kd, k::Const, a)    = leaf!(kd, 12, a, [k.σ2])
function flatten!(kd, k::Union{SumKernel,ProdKernel}, a)
    flatten!(kd, k.kleft, a); flatten!(kd, k.kright, a)
    push!(kd.ops, k isa SumKernel ? 100 : 101); push!(kd.dims_off, length(kd.dims))
end
flatten!(kd, k::Masked, a) = flatten!(kd, k.kernel, a === nothing ? collect(k.active_dims) : a[collect(k.active_dims)])
flatten!(kd, k::FixedKernel, a) = flatten!(kd, k.kernel, a)
flatten!(kd, k::Kernel, a) = throw(ArgumentError("HIPCovariance: kernel $(typeof(k)) is not on the MI355X path; use FullCovariance"))
function descriptor(k::Kernel)
    kd = KernelDesc(); flatten!(kd, k, nothing)
    isempty(kd.dims) && push!(kd.dims, 0)
    kd
end
withkernel(f, kd::KernelDesc) = GC.@preserve kd f(Ref(CKernel(length(kd.ops), pointer(kd.ops), pointer(kd.dims_off),
                                                           pointer(kd.dims), pointer(kd.params), length(kd.params))))

# ---- CovarianceStrategy + AbstractPDMat ----------------------------------------------------
# HIPCovariance()                      the dense device path: ONE n x n factor buffer (gpmi_gp_create)
# HIPCovariance(packed=true)           a BLOCKED handle on one device (gpmi_gp_create_blocked): block-rows in stripes that stop
#                                      at their own diagonal — no upper triangle, N = 250 000 in fp64 on 288 GB
# HIPCovariance(sharded=true)          with ENV["GPMI_DEVICES"] = "0,1,...": the factor row-block sharded over the GPUs of an in-process
#                                      device group — one Julia session, worker threads and peer copies inside libgpmi
# HIPCovariance(comm=rccl_comm(...))   the factor row-block sharded over the ranks of a communicator: one Julia process per GPU
#                                      (MPI.jl / Distributed launch), every rank runs the same script on the same data and gets the
#                                      same mll / alpha / predictions / gradient back
# update_mll!, update_dmll! / optimize!, predict_f (both full_cov branches), predict_y and the AbstractPDMat surface (`\`, ldiv!,
# whiten!, predict_LOO's diag(inv(cK)), cholfactors / Matrix / tr) work on all three: libgpmi answers gpmi_solve / gpmi_whiten /
# gpmi_inv_diag / gpmi_factor_to_host on blocked handles as well (results replicated on every rank; cholfactors gathers the whole
# n x n factor on the host, so Matrix / tr / unwhiten! are for sizes where that is affordable).
struct HIPCovariance <: CovarianceStrategy
    packed::Bool
    sharded::Bool         # a blocked handle without packing: on a device-group context (GPMI_DEVICES) one rank per GPU
    block::Int            # rows per block (0: library default — 1024 from 32 768 points)
    stripe_blocks::Int    # local blocks per storage stripe (packed)
    comm::Ptr{Cvoid}      # gpmi_comm* or C_NULL
end
HIPCovariance(; packed::Bool=false, sharded::Bool=false, block::Integer=0, stripe_blocks::Integer=8, comm::Ptr{Cvoid}=C_NULL) =
    HIPCovariance(packed, sharded, block, packed ? stripe_blocks : 0, comm)
blocked(s::HIPCovariance) = s.packed || s.sharded || s.comm != C_NULL

# libgpmi's own RCCL communicator (librccl opened at run time).  `exchange(id::Union{Vector{UInt8},Nothing}) -> Vector{UInt8}` moves the
# 128-byte unique id from rank 0 to every rank (MPI.bcast, a shared file, Distributed.remotecall_fetch ...).
function rccl_comm(rank::Integer, world::Integer, exchange::Function)
    id = rank == 0 ? Vector{UInt8}(undef, 128) : nothing
    if rank == 0
        ccall((:gpmi_comm_unique_id, libgpmi), Cint, (Ptr{UInt8},), id) == 0 || error("libgpmi: librccl.so could not be opened")
    end
    id = exchange(id)
    h = Ref{Ptr{Cvoid}}(C_NULL)
    check(context(), ccall((:gpmi_comm_create_rccl, libgpmi), Cint, (Ptr{Cvoid}, Ptr{UInt8}, Cint, Cint, Ptr{Ptr{Cvoid}}),
                           context(), id, rank, world, h))
    check(context(), ccall((:gpmi_comm_selftest, libgpmi), Cint, (Ptr{Cvoid}, Ptr{Cvoid}), context(), h[]))   # fail here, not inside a fit
    h[]
end
# (the generic method for ::CovarianceStrategy, src/GP.jl:12, already returns EmptyData(); spelled out so that the "nothing N x N is
#  cached" decision is visible here)
KernelData(k::Kernel, X1::AbstractMatrix, X2::AbstractMatrix, ::HIPCovariance) = EmptyData()

mutable struct HIPPDMat <: AbstractPDMat{Float64}
    handle::Ptr{Cvoid}   # gpmi_gp*
    n::Int
    xref::Any            # the x the handle was created for (re-created by update_cK! when it changes)
    xsum::UInt64         # content checksum of that x at upload time (in-place mutation of gp.x is detected, see ensure_handle!)
    strat::HIPCovariance # dense, packed or sharded (decides which create call makes the handle)
    function HIPPDMat(n, strat::HIPCovariance=HIPCovariance())
        a = new(C_NULL, n, nothing, UInt64(0), strat)
        finalizer(a) do a
            a.handle == C_NULL || ccall((:gpmi_gp_destroy, libgpmi), Cvoid, (Ptr{Cvoid},), a.handle)
        end
    end
end
alloc_cK(s::HIPCovariance, nobs) = HIPPDMat(nobs, s)                  # replaces src/GP.jl:14-20
Base.size(a::HIPPDMat) = (a.n, a.n); Base.size(a::HIPPDMat, i::Int) = a.n; dim(a::HIPPDMat) = a.n
# an AbstractPDMat is an AbstractMatrix: without these the REPL's display of a gp (or of gp.cK) would walk getindex over a matrix
# that lives on the device
Base.show(io::IO, a::HIPPDMat) = print(io, "HIPPDMat(", a.n, " x ", a.n, a.handle == C_NULL ? ", no device handle yet)" : ", factor resident on the device)")
Base.show(io::IO, ::MIME"text/plain", a::HIPPDMat) = show(io, a)
Base.getindex(a::HIPPDMat, i::Int, j::Int) = error("HIPPDMat: elementwise access is not supported (the matrix is never resident); use Matrix(a) / cholfactors(a)")

# The handle is keyed on the IDENTITY of the caller's x (gp.x, whatever its array type): the dense Float64 copy the C ABI
# needs is made only when x actually has to be uploaded, so repeated update_cK! / update_mll! calls with the same gp.x
# (every optimiser / MCMC step) reuse the resident x and buffers.  Identity alone would miss `gp.x .= ...` or an elastic
# array grown in place: the number of observations and a content checksum (O(N d), negligible next to the O(N^2) cov!)
# are compared as well, so a mutated x is re-uploaded instead of silently reusing the stale device copy.
# (Base.hash(::AbstractArray) only SAMPLES arrays of 8192+ elements, so it would miss most in-place edits at the sizes this
# library is for: every element is folded in)
xchecksum(x::AbstractMatrix) = foldl((h, v) -> hash(v, h), x; init = hash(size(x)))
function ensure_handle!(a::HIPPDMat, x::AbstractMatrix)
    if a.handle == C_NULL || a.xref !== x || size(x, 2) != a.n || xchecksum(x) != a.xsum
        a.handle == C_NULL || ccall((:gpmi_gp_destroy, libgpmi), Cvoid, (Ptr{Cvoid},), a.handle)
        a.handle = C_NULL
        xd = x isa Matrix{Float64} ? x : Matrix{Float64}(x)
        h = Ref{Ptr{Cvoid}}(C_NULL)
        rc = if blocked(a.strat)
            ccall((:gpmi_gp_create_blocked, libgpmi), Cint,
                  (Ptr{Cvoid}, Ptr{Cvoid}, Cint, Cint, Int64, Ptr{Float64}, Int64, Cint, Ptr{Ptr{Cvoid}}),
                  context(), a.strat.comm, 64, size(xd, 1), size(xd, 2), xd, a.strat.block, a.strat.stripe_blocks, h)
        else
            ccall((:gpmi_gp_create, libgpmi), Cint, (Ptr{Cvoid}, Cint, Cint, Int64, Ptr{Float64}, Ptr{Ptr{Cvoid}}),
                  context(), 64, size(xd, 1), size(xd, 2), xd, h)
        end
        check(context(), rc); a.handle = h[]; a.xref = x; a.n = size(xd, 2); a.xsum = xchecksum(x)
    end
    a
end

# (not called fit!: GaussianProcesses exports its own fit!(gp, x, y), src/GPE.jl:128, and a second function of that name in this module
#  would shadow or — once the exported one has been referenced — refuse to be defined)
function device_fit!(a::HIPPDMat, x, kernel, logNoise, ymμ::Vector{Float64}, alpha::Union{Vector{Float64},Nothing})
    ensure_handle!(a, x)
    ln = logNoise isa Real ? Float64[logNoise] : Vector{Float64}(logNoise)
    mll = Ref{Float64}(NaN); info = Ref{Int64}(0)
    rc = withkernel(descriptor(kernel)) do ck
        ccall((:gpmi_fit, libgpmi), Cint,
              (Ptr{Cvoid}, Ref{CKernel}, Ptr{Float64}, Int64, Ptr{Float64}, Ref{Float64}, Ptr{Float64}, Ref{Int64}),
              a.handle, ck, ln, length(ln), ymμ, mll, alpha === nothing ? C_NULL : alpha, info)
    end
    check(context(), rc, info[])
    mll[]
end

# update_cK! alone (src/GPE.jl:169-195; callers: update_cK!(gp) :193-195, GPA with logNoise = -20, ElasticGPE): factor only.
# TWO methods typed like the reference's two (logNoise::Real :169, ::AbstractVector :177): an untyped logNoise would be ambiguous with both
# (neither signature more specific), and update_cK!(gp) would throw a MethodError.
function update_cK!(cK::HIPPDMat, x::AbstractMatrix, kernel::Kernel, logNoise::Real, data::KernelData, ::HIPCovariance)
    device_fit!(cK, x, kernel, logNoise, zeros(size(x, 2)), nothing)
    cK
end
function update_cK!(cK::HIPPDMat, x::AbstractMatrix, kernel::Kernel, logNoise::AbstractVector, data::KernelData, ::HIPCovariance)
    device_fit!(cK, x, kernel, logNoise, zeros(size(x, 2)), nothing)
    cK
end

# update_mll! (src/GPE.jl:202-212) fused: cov! + nugget + Cholesky + alpha + logdet + mll in one device pass.
# gp.alpha is #undef when the inner constructor's initialise_target! makes the first call (src/GPE.jl:42-44 leaves the fields after cK
# unassigned; the reference's own update_mll! only ever ASSIGNS gp.alpha), so it is tested with isdefined before it is read.
function update_mll!(gp::GPE{X,Y,M,K,HIPCovariance}; noise::Bool=true, domean::Bool=true, kern::Bool=true) where {X,Y,M,K}
    μ = mean(gp.mean, gp.x)
    ymμ = Vector{Float64}(gp.y - μ)
    (isdefined(gp, :alpha) && length(gp.alpha) == gp.nobs) || (gp.alpha = Vector{Float64}(undef, gp.nobs))
    if kern | noise
        gp.mll = device_fit!(gp.cK, gp.x, gp.kernel, get_value(gp.logNoise), ymμ, gp.alpha)
    else
        # only the mean changed (src/GPE.jl:203-211 with update_cK! skipped): the factor is kept, alpha = cK \ (y - mu), and the DEVICE copy of
        # alpha — what gpmi_predict and gpmi_grad read — is replaced by the same call
        mll = Ref{Float64}(NaN)
        check(context(), ccall((:gpmi_update_alpha, libgpmi), Cint, (Ptr{Cvoid}, Ptr{Float64}, Ref{Float64}, Ptr{Float64}),
                               gp.cK.handle, ymμ, mll, gp.alpha))
        gp.mll = mll[]
    end
    gp
end

# slots of the free parameters inside the kernel tree's full parameter list (FixedKernel hides some: fixed_kernel.jl:63-66)
full_slots(k::Kernel) = collect(1:num_params(k))
full_slots(k::FixedKernel) = full_slots(k.kernel)[k.free]
full_slots(k::Masked) = full_slots(k.kernel)
full_slots(k::Union{SumKernel,ProdKernel}) = [full_slots(k.kleft); full_nparams(k.kleft) .+ full_slots(k.kright)]
full_nparams(k::Kernel) = num_params(k)
full_nparams(k::Union{FixedKernel,Masked}) = full_nparams(k.kernel)
full_nparams(k::Union{SumKernel,ProdKernel}) = full_nparams(k.kleft) + full_nparams(k.kright)
# The gradient goes through the reference's own SEAM (src/GPE.jl:245-324), not around it: optimize!, mcmc and vi call
# update_dmll!(gp, precomp) / update_target_and_dtarget!(gp, precomp) with precomp = init_precompute(gp) (src/GPE.jl:260,298,382-391,
# src/mcmc.jl:10-17), and the generic update_dmll! then asks precompute!, dmll_noise, dmll_mean!, dmll_kern! in turn.  For the exact
# path the reference's precompute is the N x N matrix alpha alpha' - K^-1 (FullCovariancePrecompute, get_ααinvcKI!) that dmll_kern!
# re-walks once per parameter; here precompute! IS the whole device pass (ONE gpmi_grad call: K^-1 block by block, the fused dK/dtheta
# trace for all parameters, the noise trace) and dmll_kern! / dmll_noise hand out its results.  dmll_mean! stays the generic one.
struct HIPGradientPrecompute <: AbstractGradientPrecompute
    dkern::Vector{Float64}            # d mll / d theta_p for every parameter of the kernel TREE (fixed ones included), get_params order
    dnoise::Base.RefValue{Float64}
end
init_precompute(::HIPCovariance, X, y, k::Kernel) = HIPGradientPrecompute(Vector{Float64}(undef, max(full_nparams(k), 1)), Ref(0.0))
function precompute!(p::HIPGradientPrecompute, gp::GPBase)
    nfull = full_nparams(gp.kernel)
    length(p.dkern) >= max(nfull, 1) || resize!(p.dkern, max(nfull, 1))
    # a heteroscedastic model (VectorParam logNoise) has kernel and mean gradients but no noise gradient — in the reference too
    # (src/GPE.jl:276-278 is commented out, update_dmll! asserts num_params(gp.logNoise) == 1 when noise = true, :313): the vector is
    # passed through and the noise output is not requested
    lnv = get_value(gp.logNoise)
    ln = lnv isa Real ? Float64[lnv] : Vector{Float64}(lnv)
    p.dnoise[] = NaN
    rc = withkernel(descriptor(gp.kernel)) do ck
        ccall((:gpmi_grad, libgpmi), Cint, (Ptr{Cvoid}, Ref{CKernel}, Ptr{Float64}, Int64, Ptr{Float64}, Int32, Ptr{Float64}),
              gp.cK.handle, ck, ln, length(ln), p.dkern, nfull, length(ln) == 1 ? p.dnoise : Ptr{Float64}(C_NULL))
    end
    check(context(), rc)
    p
end
function dmll_kern!(dmll::AbstractVector, gp::GPBase, p::HIPGradientPrecompute, ::HIPCovariance)       # src/GPE.jl:265-267
    dmll .= @view p.dkern[full_slots(gp.kernel)]
    dmll
end
dmll_noise(gp::GPE, p::HIPGradientPrecompute, ::HIPCovariance) = p.dnoise[]                               # src/GPE.jl:279-281

function \(a::HIPPDMat, b::DenseVecOrMat{Float64})                      # PDMats `\`  (src/GPE.jl:208)
    out = copy(b)
    check(context(), ccall((:gpmi_solve, libgpmi), Cint, (Ptr{Cvoid}, Int64, Ptr{Float64}), a.handle, size(out, 2), out)); out
end
\(a::HIPPDMat, b::AbstractVecOrMat{<:Real}) = a \ Array{Float64}(b)     # views, adjoints, integer right-hand sides
ldiv!(a::HIPPDMat, b::DenseVecOrMat{Float64}) =
    (check(context(), ccall((:gpmi_solve, libgpmi), Cint, (Ptr{Cvoid}, Int64, Ptr{Float64}), a.handle, size(b, 2), b)); b)
whiten!(a::HIPPDMat, b::DenseVecOrMat{Float64}) =                       # src/GP.jl:27
    (check(context(), ccall((:gpmi_whiten, libgpmi), Cint, (Ptr{Cvoid}, Int64, Ptr{Float64}), a.handle, size(b, 2), b)); b)
whiten(a::HIPPDMat, b::DenseVecOrMat{Float64}) = whiten!(a, copy(b))
function logdet(a::HIPPDMat)                                            # src/GPE.jl:210
    out = Ref{Float64}(NaN)
    check(context(), ccall((:gpmi_logdet, libgpmi), Cint, (Ptr{Cvoid}, Ref{Float64}), a.handle, out)); out[]
end
function cholfactors(a::HIPPDMat)                                       # src/GP.jl:89 (lazy host copy)
    U = Matrix{Float64}(undef, a.n, a.n)
    check(context(), ccall((:gpmi_factor_to_host, libgpmi), Cint, (Ptr{Cvoid}, Ptr{Float64}), a.handle, U)); U
end
# predict_LOO(Σ, alpha, y) (src/crossvalidation.jl:8-13) needs only diag(inv(Σ)): n^3/3 on the device instead of inv(Σ)
function predict_LOO(a::HIPPDMat, alpha::AbstractVector{<:Real}, y::AbstractVector{<:Real})
    d = Vector{Float64}(undef, a.n)
    check(context(), ccall((:gpmi_inv_diag, libgpmi), Cint, (Ptr{Cvoid}, Ptr{Float64}), a.handle, d))
    σi2 = 1 ./ d
    return -alpha .* σi2 .+ y, σi2
end
Base.Matrix(a::HIPPDMat) = (U = UpperTriangular(cholfactors(a)); Matrix(U' * U))
# unwhiten!(a, x) = L x with a = L Lᵀ (PDMats; the reference's only call is rand!, src/GP.jl:136, on the P x P predictive
# covariance — a plain PDMat — so this method is for completeness of the AbstractPDMat surface: host product with the
# fetched factor).  tr(a) as SubsetOfRegsPDMat defines it (subsetofregressors.jl:61-72): trace of the covariance itself,
# = ‖U‖²_F.
unwhiten!(a::HIPPDMat, b::DenseVecOrMat{Float64}) = lmul!(UpperTriangular(cholfactors(a))', b)
tr(a::HIPPDMat) = sum(abs2, UpperTriangular(cholfactors(a)))
mat(a::HIPPDMat) = Matrix(a)                                            # K is regenerated on demand, never resident

# ---- predict: batches the full_cov=false branch (src/GP.jl:69-77 is P separate trsv) -----------
function hip_predict(cK::HIPPDMat, kernel::Kernel, meanf::Mean, x::AbstractMatrix, full_cov::Bool)
    xp = Matrix{Float64}(x); P = size(xp, 2)
    mx = Vector{Float64}(mean(meanf, xp)); μ = Vector{Float64}(undef, P)
    Σ = full_cov ? Matrix{Float64}(undef, P, P) : Vector{Float64}(undef, P)
    rc = withkernel(descriptor(kernel)) do ck
        ccall((:gpmi_predict, libgpmi), Cint,
              (Ptr{Cvoid}, Ref{CKernel}, Int64, Ptr{Float64}, Ptr{Float64}, Cint, Ptr{Float64}, Ptr{Float64}),
              cK.handle, ck, P, xp, mx, full_cov ? 1 : 0, μ, Σ)
    end
    check(context(), rc); μ, Σ
end
# Element types the device path takes (marshalled to Float64 at the C boundary).  ANY OTHER eltype — ForwardDiff.Dual test points pushed
# through predict_y (test/kernels.jl:174-181), BigFloat, ... — keeps the reference's eltype-generic path (src/GP.jl:64-79 allocates
# `Array{eltype(x)}`): no predict_f method is defined for it here, so dispatch lands on the reference's predict_f(::GPBase, ::AbstractMatrix),
# which reaches predictMVN below one point at a time, and THAT method runs the reference's own host algebra (src/GP.jl:25-49) against a
# host copy of the factor.  (Round 5's single method accepted every eltype and then threw in `Matrix{Float64}(x)`.)
const DeviceReal = Union{AbstractFloat, Integer}
device_eltype(x::AbstractMatrix) = eltype(x) <: DeviceReal && !(eltype(x) <: BigFloat)
host_pdmat(a::HIPPDMat) = PDMat(Cholesky(UpperTriangular(cholfactors(a))))    # K = UᵀU: the reference's own PDMat (src/GP.jl:16-17), on the host
function predict_f(gp::GPE{X,Y,M,K,HIPCovariance}, x::AbstractMatrix{<:DeviceReal}; full_cov::Bool=false) where {X,Y,M,K}
    size(x, 1) == gp.dim || throw(ArgumentError("Gaussian Process object and input observations do not have consistent dimensions"))
    device_eltype(x) || return invoke(predict_f, Tuple{GPBase, AbstractMatrix}, gp, x; full_cov=full_cov)
    hip_predict(gp.cK, gp.kernel, gp.mean, x, full_cov)
end
# predictMVN (src/GP.jl:39-49; reached by predict_full, src/GPE.jl:399, for callers that bypass predict_f — and by the reference's
# generic predict_f for the eltypes the device does not take): the full predictive covariance from the device, or the reference's generic
# method on a host PDMat.  `alpha` is the one the last gpmi_fit left in the handle (= gp.alpha).
function predictMVN(xpred::AbstractMatrix, xtrain::AbstractMatrix, ytrain::AbstractVector, kernel::Kernel, meanf::Mean,
                    alpha::AbstractVector, ::HIPCovariance, Ktrain::HIPPDMat)
    device_eltype(xpred) && return hip_predict(Ktrain, kernel, meanf, xpred, true)
    predictMVN(xpred, xtrain, ytrain, kernel, meanf, alpha, GaussianProcesses.FullCovariance(), host_pdmat(Ktrain))
end

# convenience constructor, as SoR(...)/FITC(...) are (src/sparse/subsetofregressors.jl:324-327)
GP_hip(x::AbstractMatrix, y::AbstractVector, m::Mean, k::Kernel, logNoise=-2.0; kw...) = GPE(x, y, m, k, logNoise, HIPCovariance(; kw...))

# ---- FITC (src/sparse/fully_indep_train_conditional.jl) on the device ------------------------------------
# HIPFITC plays FullyIndepStrat's role (:111-113); HIPFITCPDMat the FullyIndepPDMat's (:8-19): the n x m matrices,
# Lambda and both factors stay in HBM (gpmi_fitc_*), only alpha / mll / predictions cross the bus.
struct HIPFITC{M<:AbstractMatrix} <: SparseStrategy
    inducing::M
end
mutable struct HIPFITCPDMat <: SparsePDMat{Float64}
    handle::Ptr{Cvoid}   # gpmi_fitc*
    n::Int
    inducing::Matrix{Float64}
    xref::Any
    xsum::UInt64
    function HIPFITCPDMat(n, inducing)
        a = new(C_NULL, n, Matrix{Float64}(inducing), nothing, UInt64(0))
        finalizer(a) do a
            a.handle == C_NULL || ccall((:gpmi_fitc_destroy, libgpmi), Cvoid, (Ptr{Cvoid},), a.handle)
        end
    end
end
alloc_cK(s::HIPFITC, nobs) = HIPFITCPDMat(nobs, s.inducing)             # replaces fully_indep…:118-132
# more specific than KernelData(k, X1, X2, ::SparseStrategy) (src/sparse/sparsekerneldata.jl:18), which would build three host distance caches
KernelData(k::Kernel, X1::AbstractMatrix, X2::AbstractMatrix, ::HIPFITC) = EmptyData()
Base.size(a::HIPFITCPDMat) = (a.n, a.n); Base.size(a::HIPFITCPDMat, i::Int) = a.n; dim(a::HIPFITCPDMat) = a.n
Base.show(io::IO, a::HIPFITCPDMat) = print(io, "HIPFITCPDMat(n = ", a.n, ", m = ", size(a.inducing, 2), ")")
Base.show(io::IO, ::MIME"text/plain", a::HIPFITCPDMat) = show(io, a)
Base.getindex(a::HIPFITCPDMat, i::Int, j::Int) = error("HIPFITCPDMat: elementwise access is not supported (the n x m factors live on the device)")
function ensure_handle!(a::HIPFITCPDMat, x::AbstractMatrix)
    if a.handle == C_NULL || a.xref !== x || size(x, 2) != a.n || xchecksum(x) != a.xsum
        a.handle == C_NULL || ccall((:gpmi_fitc_destroy, libgpmi), Cvoid, (Ptr{Cvoid},), a.handle)
        a.handle = C_NULL
        xd = x isa Matrix{Float64} ? x : Matrix{Float64}(x)
        h = Ref{Ptr{Cvoid}}(C_NULL)
        rc = ccall((:gpmi_fitc_create, libgpmi), Cint,
                   (Ptr{Cvoid}, Cint, Cint, Int64, Ptr{Float64}, Int64, Ptr{Float64}, Ptr{Ptr{Cvoid}}),
                   context(), 64, size(xd, 1), size(xd, 2), xd, size(a.inducing, 2), a.inducing, h)
        check(context(), rc); a.handle = h[]; a.xref = x; a.n = size(xd, 2); a.xsum = xchecksum(x)
    end
    a
end
# update_mll! (src/GPE.jl:202-212) over update_cK!(::FullyIndepPDMat) (:134-156), its `\` (:38-41) and logdet (:80)
function fitc_fit!(a::HIPFITCPDMat, x, kernel, logNoise::Real, ymμ::Vector{Float64}, alpha::Union{Vector{Float64},Nothing})
    ensure_handle!(a, x)
    mll = Ref{Float64}(NaN); info = Ref{Int64}(0)
    rc = withkernel(descriptor(kernel)) do ck
        ccall((:gpmi_fitc_fit, libgpmi), Cint,
              (Ptr{Cvoid}, Ref{CKernel}, Float64, Ptr{Float64}, Ref{Float64}, Ptr{Float64}, Ref{Int64}),
              a.handle, ck, Float64(logNoise), ymμ, mll, alpha === nothing ? Ptr{Float64}(C_NULL) : alpha, info)
    end
    check(context(), rc, info[]); mll[]
end
function update_mll!(gp::GPE{X,Y,M,K,<:HIPFITC}; noise::Bool=true, domean::Bool=true, kern::Bool=true) where {X,Y,M,K}
    lnv = get_value(gp.logNoise)
    lnv isa Real || throw(ArgumentError("HIPFITC takes a scalar logNoise (as FITC(...) does, fully_indep_train_conditional.jl:333)"))
    ymμ = Vector{Float64}(gp.y - mean(gp.mean, gp.x))
    (isdefined(gp, :alpha) && length(gp.alpha) == gp.nobs) || (gp.alpha = Vector{Float64}(undef, gp.nobs))
    gp.mll = fitc_fit!(gp.cK, gp.x, gp.kernel, lnv, ymμ, gp.alpha)   # (a mean-only update refits: 2nm^2, no kept-factor shortcut)
    gp
end
# update_cK!(gp) (src/GPE.jl:193-195) on a FITC model: without this method the call would fall to the exact-path update_cK! (:169), which asks
# mat(cK) of a matrix that does not exist
function update_cK!(cK::HIPFITCPDMat, x::AbstractMatrix, kernel::Kernel, logNoise::Real, data::KernelData, ::HIPFITC)
    fitc_fit!(cK, x, kernel, logNoise, zeros(size(x, 2)), nothing)
    cK
end
# predictMVN(::FullyIndepStrat) (:321-329 -> determ_train_conditional.jl:41-59 -> subsetofregressors.jl:303-321)
function fitc_predict(cK::HIPFITCPDMat, kernel::Kernel, meanf::Mean, x::AbstractMatrix, full_cov::Bool)
    xp = Matrix{Float64}(x); P = size(xp, 2)
    mx = Vector{Float64}(mean(meanf, xp)); μ = Vector{Float64}(undef, P)
    Σ = full_cov ? Matrix{Float64}(undef, P, P) : Vector{Float64}(undef, P)
    rc = withkernel(descriptor(kernel)) do ck
        ccall((:gpmi_fitc_predict, libgpmi), Cint,
              (Ptr{Cvoid}, Ref{CKernel}, Int64, Ptr{Float64}, Ptr{Float64}, Cint, Ptr{Float64}, Ptr{Float64}),
              cK.handle, ck, P, xp, mx, full_cov ? 1 : 0, μ, Σ)
    end
    check(context(), rc); μ, Σ
end
function predict_f(gp::GPE{X,Y,M,K,<:HIPFITC}, x::AbstractMatrix; full_cov::Bool=false) where {X,Y,M,K}
    size(x, 1) == gp.dim || throw(ArgumentError("Gaussian Process object and input observations do not have consistent dimensions"))
    fitc_predict(gp.cK, gp.kernel, gp.mean, x, full_cov)
end
# ... and for callers of predict_full (src/GPE.jl:399), which bypasses predict_f: without this method the generic predictMVN (src/GP.jl:39-49)
# would be chosen and ask whiten! of the device-side factor
predictMVN(xpred::AbstractMatrix, xtrain::AbstractMatrix, ytrain::AbstractVector, kernel::Kernel, meanf::Mean,
           alpha::AbstractVector, ::HIPFITC, Ktrain::HIPFITCPDMat) = fitc_predict(Ktrain, kernel, meanf, xpred, true)
# the same seam for the FITC strategy: precompute! = gpmi_fitc_grad (dmll_kern!(…, ::FullyIndepStrat) fully_indep…:200-234 over
# subsetofregressors.jl:219-256, dmll_noise :243-257, precompute! subsetofregressors.jl:141-151 — all of it in one device pass)
init_precompute(::HIPFITC, X, y, k::Kernel) = HIPGradientPrecompute(Vector{Float64}(undef, max(full_nparams(k), 1)), Ref(0.0))
function precompute!(p::HIPGradientPrecompute, gp::GPE{X,Y,M,K,<:HIPFITC}) where {X,Y,M,K}   # more specific than (p, ::GPBase) above
    nfull = full_nparams(gp.kernel)
    length(p.dkern) >= max(nfull, 1) || resize!(p.dkern, max(nfull, 1))
    rc = withkernel(descriptor(gp.kernel)) do ck
        ccall((:gpmi_fitc_grad, libgpmi), Cint, (Ptr{Cvoid}, Ref{CKernel}, Float64, Ptr{Float64}, Int32, Ref{Float64}),
              gp.cK.handle, ck, Float64(get_value(gp.logNoise)), p.dkern, nfull, p.dnoise)
    end
    check(context(), rc)
    p
end
dmll_kern!(dmll::AbstractVector, gp::GPBase, p::HIPGradientPrecompute, ::HIPFITC) = (dmll .= @view p.dkern[full_slots(gp.kernel)]; dmll)
dmll_noise(gp::GPE, p::HIPGradientPrecompute, ::HIPFITC) = p.dnoise[]
function get_alpha_u(a::HIPFITCPDMat, args...)                           # fully_indep…:279-286
    au = Vector{Float64}(undef, size(a.inducing, 2))
    check(context(), ccall((:gpmi_fitc_alpha_u, libgpmi), Cint, (Ptr{Cvoid}, Ptr{Float64}), a.handle, au)); au
end
FITC_hip(x::AbstractMatrix, inducing::AbstractMatrix, y::AbstractVector, m::Mean, k::Kernel, logNoise::Real) =
    GPE(x, y, m, k, logNoise, HIPFITC(inducing))                        # as FITC(...) :333-336

export HIPCovariance, HIPPDMat, GP_hip, rccl_comm, HIPFITC, HIPFITCPDMat, FITC_hip
end # module
