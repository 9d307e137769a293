# runtests.jl — what a GaussianProcesses.jl maintainer runs on day one, on a box with an MI355X:
#
#     LIBGPMI=/path/to/gaussianprocesses.jl_amd/lib/libgpmi.so julia --project=<env with GaussianProcesses> runtests.jl
#
# STATUS: like GPMI355X.jl, written in a container without Julia and never executed (INTEGRATION.md).  Every check below has an
# executed twin in the Python suite (tests/test_gpu_parity.py, tests/test_reference_goldens.py, tests/test_gpu_fitc.py) that drives the
# same C entry points; this file checks the one thing those cannot: that the Julia methods of GPMI355X.jl are SELECTED by dispatch where
# the reference's callers reach them (update_cK!(gp), optimize!, predict_y, rand, predict_LOO, FITC), with no ambiguity and no MethodError.
#
# Mirrors the reference's own tests: test/gp.jl:20-86 (constructors, predictions at the observations, update after mutating the
# kernel, rand, parameter round trip), test/kernels.jl:148-172 + 209-245 (the kernel list, cov / gradient consistency — here as
# HIPCovariance against FullCovariance on the same data), test/optim.jl:20-37 (optimize!, fixed kernel), test/heteroscedastic.jl,
# test/test_sparse.jl:156 (FITC's stored value), docs/src/Regression.md:60-63,83-89,118-124 (printed transcript).
using Test, Random, LinearAlgebra, Statistics, PDMats
import ForwardDiff
using GaussianProcesses
using GaussianProcesses: get_params, set_params!, update_target!, update_target_and_dtarget!, update_mll!, update_cK!, init_precompute,
    predict_LOO, fix, get_param_names
include(joinpath(@__DIR__, "GPMI355X.jl"))
using .GPMI355X

const RTOL = 1e-8      # fp64 device path against the reference's own CPU path on identical inputs (north_star asks for 1e-5)

# the kernels on the MI355X path, as test/kernels.jl:209-245 lists them (d = 3)
function kernel_list(d)
    ll = collect(range(-0.3, 0.4; length=d))
    [SEIso(0.3, 0.1), SEArd(ll, 0.1), Mat12Iso(0.2, -0.1), Mat12Ard(ll, -0.1), Mat32Iso(0.2, 0.4), Mat32Ard(ll, 0.4),
     Mat52Iso(-0.1, 0.2), Mat52Ard(ll, 0.2), RQIso(0.3, 0.2, 0.4), RQArd(ll, 0.2, 0.4),
     SEIso(0.3, 0.1) + RQIso(0.3, 0.2, 0.4), SEIso(0.3, 0.1) * Mat12Iso(0.2, -0.1),
     SEArd(ll, 0.0) + Mat52Iso(log(0.7), log(0.5)) + Noise(log(0.05)),
     (SEIso(0.3, 0.3) + Mat12Iso(0.3, 0.3)) * RQIso(0.3, 0.3, 0.3),
     Masked(SEIso(0.3, 0.3), [1]), Masked(SEIso(0.3, 0.3), [1]) + Masked(RQArd([0.1, 0.2], 0.3, 0.3), [2, 3]),
     fix(SEIso(0.3, 0.3), :lσ), Const(0.2) * Masked(Mat32Ard([0.4, -0.3], 0.0), [3, 1])]
end

both(x, y, m, k, ln) = (GPE(x, y, deepcopy(m), deepcopy(k), ln), GPE(x, y, deepcopy(m), deepcopy(k), ln, HIPCovariance()))

@testset "GPMI355X" begin
    Random.seed!(1)
    d, n = 3, 200
    X = 2π * rand(d, n)
    y = [sum(sin, view(X, :, i)) / d for i in 1:n] + 0.05 * randn(n)
    Xtest = 2π * rand(d, 37)

    @testset "constructors (test/gp.jl:20-25)" begin
        k = SE(0.0, 0.0)
        @test GP_hip(X, y, MeanZero(), k) isa GPE
        @test GPE(X, y, MeanZero(), k, 1.2, HIPCovariance()) isa GPE
        @test GPE(X, y, MeanZero(), k, GaussianProcesses.Scalar(1.2), HIPCovariance()) isa GPE
        gp = GPE(X, y, MeanZero(), k, -1.0, HIPCovariance())
        @test gp.cK isa HIPPDMat
        @test gp.data isa GaussianProcesses.EmptyData
        @test sprint(show, gp.cK) isa String                     # an AbstractMatrix whose display must not walk getindex
        @test_throws ArgumentError GPE(X, y, MeanZero(), LinIso(0.0), -1.0, HIPCovariance())   # not on the device path: descriptor refuses
        @test_throws ArgumentError GPE(X, y[1:end-1], MeanZero(), k, -1.0, HIPCovariance())    # src/GPE.jl:42
    end

    @testset "update_mll!, alpha, predict_f (both branches), predict_y vs FullCovariance: $(typeof(k).name.name) #$i" for (i, k) in enumerate(kernel_list(d))
        cpu, hip = both(X, y, MeanConst(0.1), k, log(0.1))
        @test hip.mll ≈ cpu.mll rtol = RTOL
        @test hip.alpha ≈ cpu.alpha rtol = 1e-6
        @test logdet(hip.cK) ≈ logdet(cpu.cK) rtol = RTOL
        μc, σc = predict_f(cpu, Xtest); μh, σh = predict_f(hip, Xtest)
        @test μh ≈ μc rtol = 1e-6
        @test σh ≈ σc rtol = 1e-5 atol = 1e-9
        μc, Σc = predict_f(cpu, Xtest; full_cov=true); μh, Σh = predict_f(hip, Xtest; full_cov=true)
        @test μh ≈ μc rtol = 1e-6
        @test Σh ≈ Σc rtol = 1e-5 atol = 1e-9
        yc, vc = predict_y(cpu, Xtest); yh, vh = predict_y(hip, Xtest)          # src/GPE.jl:408-416 on top of predict_f
        @test vh ≈ vc rtol = 1e-5 atol = 1e-9
        μf, Σf = GaussianProcesses.predict_full(hip, Xtest)                     # predictMVN(…, ::HIPCovariance, ::HIPPDMat)
        @test Σf ≈ Σc rtol = 1e-5 atol = 1e-9
    end

    @testset "predictions at the observations (test/gp.jl:35-42)" begin
        gp = GPE(X, y, MeanZero(), SE(0.0, 0.0), -2.0, HIPCovariance())
        y_pred, σ2 = predict_y(gp, X)
        @test maximum(abs, gp.y - y_pred) ≈ 0.0 atol = 0.1
        y_pred, pred_cov = predict_y(gp, X; full_cov=true)
        @test σ2 ≈ diag(pred_cov) rtol = 1e-6 atol = 1e-9
    end

    @testset "one-dimensional inputs (x::AbstractVector, src/GPE.jl:96-97)" begin
        x1 = 2π * rand(40); y1 = sin.(x1) + 0.05 * randn(40)
        cpu = GPE(x1, y1, MeanZero(), SE(0.0, 0.0), -1.0)
        hip = GPE(x1', y1, MeanZero(), SE(0.0, 0.0), -1.0, HIPCovariance())       # x' is an Adjoint: ensure_handle! makes the dense copy
        @test hip.mll ≈ cpu.mll rtol = RTOL
        xs = collect(range(0, stop=2π, length=25))
        @test predict_y(hip, xs)[1] ≈ predict_y(cpu, xs)[1] rtol = 1e-6           # predict_y(gp, x::AbstractVector) -> x'
    end

    @testset "update after mutating the kernel (test/gp.jl:67-73), params round trip (:82-87)" begin
        cpu, hip = both(X, y, MeanZero(), SE(0.0, 0.0), -1.0)
        cpu.kernel.ℓ2 = 4.0; hip.kernel.ℓ2 = 4.0
        update_target!(cpu); update_target!(hip)
        @test hip.target ≈ cpu.target rtol = RTOL
        p1 = deepcopy(get_params(hip)); set_params!(hip, p1); @test get_params(hip) ≈ p1
        set_params!(hip, p1 .+ 0.1); set_params!(cpu, p1 .+ 0.1); update_target!(hip); update_target!(cpu)
        @test hip.mll ≈ cpu.mll rtol = RTOL
    end

    @testset "update_cK!(gp): the reference's two logNoise methods, no ambiguity (src/GPE.jl:169,177,193)" begin
        _, hip = both(X, y, MeanZero(), SEArd(zeros(d), 0.0), -1.0)
        @test update_cK!(hip) === hip.cK
        @test isempty(detect_ambiguities(GPMI355X, GaussianProcesses))            # every method of the shim against the reference's
        lnv = collect(range(-2.0, -0.5; length=n))
        cpu = GPE(X, y, MeanZero(), SEIso(-0.5, 0.0), lnv)
        het = GPE(X, y, MeanZero(), SEIso(-0.5, 0.0), lnv, HIPCovariance())       # VectorParam: update_cK!(…, logNoise::AbstractVector, …)
        @test het.mll ≈ cpu.mll rtol = RTOL
        @test update_cK!(het) === het.cK
        pre = init_precompute(het)
        GaussianProcesses.update_dmll!(het, pre; noise=false)                       # kernel / mean gradient of a heteroscedastic model
        prc = init_precompute(cpu); GaussianProcesses.update_dmll!(cpu, prc; noise=false)
        @test het.dmll ≈ cpu.dmll rtol = 1e-6
    end

    @testset "mean-only update keeps the factor and replaces the device alpha (src/GPE.jl:203-211)" begin
        cpu, hip = both(X, y, MeanLin([0.1, -0.2, 0.3]), SEArd(zeros(d), 0.0), -1.0)
        for g in (cpu, hip)
            set_params!(g.mean, [0.5, 0.1, -0.4]); update_mll!(g; kern=false, noise=false)
        end
        @test hip.mll ≈ cpu.mll rtol = RTOL
        @test predict_f(hip, Xtest)[1] ≈ predict_f(cpu, Xtest)[1] rtol = 1e-6     # reads the device alpha
        update_target_and_dtarget!(cpu); pre = init_precompute(hip); GaussianProcesses.update_dmll!(hip, pre)
        @test hip.dmll ≈ cpu.dmll rtol = 1e-6
    end

    @testset "gradient through the reference's seam: $(typeof(k).name.name) #$i" for (i, k) in enumerate(kernel_list(d))
        cpu, hip = both(X, y, MeanConst(0.1), k, log(0.1))
        update_target_and_dtarget!(cpu); update_target_and_dtarget!(hip)          # init_precompute -> precompute! -> dmll_noise / dmll_mean! / dmll_kern!
        @test length(hip.dtarget) == length(cpu.dtarget)
        @test hip.dtarget ≈ cpu.dtarget rtol = 1e-6 atol = 1e-8 * maximum(abs, cpu.dtarget)
        update_target_and_dtarget!(hip; noise=false, domean=false)
        update_target_and_dtarget!(cpu; noise=false, domean=false)
        @test hip.dtarget ≈ cpu.dtarget rtol = 1e-6 atol = 1e-8 * maximum(abs, cpu.dtarget)
    end

    @testset "optimize! (test/optim.jl:20-37)" begin
        Xo = rand(2, 20); yo = Xo'rand(2) .+ sin.(Xo[1, :] * 2π) .+ 0.1 * randn(20)
        gp = GPE(Xo, yo, MeanLin(zeros(2)), SE(log(0.5), 1.0), -3.0, HIPCovariance())
        init_target = gp.target
        optimize!(gp)
        @test gp.target > init_target
        kern = SE(log(0.5), 1.0)
        init_param = get_params(kern)[1]
        gpf = GPE(Xo, yo, MeanZero(), fix(deepcopy(kern), get_param_names(kern)[1]), -1.0, HIPCovariance())
        t0 = gpf.target; optimize!(gpf)
        @test gpf.target > t0
        @test get_params(kern)[1] == init_param
    end

    @testset "PosDefException(info) and recovery (src/GP.jl:110, src/optimize.jl:56-58)" begin
        Xd = hcat(X[:, 1:50], X[:, 1:50])                                         # duplicated points, almost no noise
        yd = vcat(y[1:50], y[1:50])
        @test_throws PosDefException GPE(Xd, yd, MeanZero(), SEIso(0.0, 0.0), -20.0, HIPCovariance())
        gp = GPE(Xd, yd, MeanZero(), SEIso(0.0, 0.0), -1.0, HIPCovariance())
        m0 = gp.mll
        set_params!(gp, [-20.0, 0.0, 0.0])
        err = try update_target!(gp); nothing catch e; e end
        @test err isa PosDefException && err.info > 0
        set_params!(gp, [-1.0, 0.0, 0.0]); update_target!(gp)                     # the handle survives the failed factorisation
        @test gp.mll ≈ m0 rtol = 1e-12
    end

    @testset "rand, predict_LOO, the AbstractPDMat surface" begin
        cpu, hip = both(X, y, MeanZero(), SEArd(zeros(d), 0.0), -1.0)
        @test size(rand(hip, Xtest, 3)) == (size(Xtest, 2), 3)                     # src/GP.jl:120-146
        μc, σc = predict_LOO(cpu); μh, σh = predict_LOO(hip)                       # src/crossvalidation.jl:8-13
        @test μh ≈ μc rtol = 1e-6
        @test σh ≈ σc rtol = 1e-6
        B = randn(n, 3)
        @test hip.cK \ B ≈ cpu.cK \ B rtol = 1e-6
        @test hip.cK \ view(B, :, 1) ≈ cpu.cK \ B[:, 1] rtol = 1e-6
        @test whiten(hip.cK, B) ≈ whiten(cpu.cK, B) rtol = 1e-6
        @test UpperTriangular(GaussianProcesses.cholfactors(hip.cK)) ≈ UpperTriangular(GaussianProcesses.cholfactors(cpu.cK)) rtol = 1e-8
        @test Matrix(hip.cK) ≈ Matrix(cpu.cK) rtol = 1e-8
    end

    @testset "eltype-generic predict (src/GP.jl:70; test/kernels.jl:174-181: ForwardDiff through predict_y)" begin
        # Dual-valued test points are not a device eltype: dispatch must fall to the reference's predict_f(::GPBase, ::AbstractMatrix) and from
        # there to the shim's predictMVN, which runs the reference's host algebra on a host copy of the factor — same numbers, derivatives intact
        cpu, hip = both(X, y, MeanConst(0.0), SEArd(zeros(d), 0.0), -3.0)
        f_hip = z -> sum(predict_y(hip, reshape(z, :, 1)))[1]
        f_cpu = z -> sum(predict_y(cpu, reshape(z, :, 1)))[1]
        z = rand(d)
        @test f_hip(z) ≈ f_cpu(z) rtol = RTOL                                       # Float64: the device path
        @test ForwardDiff.gradient(f_hip, z) ≈ ForwardDiff.gradient(f_cpu, z) rtol = 1e-8   # (ForwardDiff: a dependency of GaussianProcesses itself, Project.toml:12)
        μb, σb = predict_f(hip, big.(Xtest[:, 1:3]))                                 # BigFloat: also the generic path
        μc, σc = predict_f(cpu, Xtest[:, 1:3])
        @test Float64.(μb) ≈ μc rtol = 1e-8
        @test Float64.(σb) ≈ σc rtol = 1e-6
    end

    @testset "packed storage on one device (gpmi_gp_create_blocked)" begin
        cpu = GPE(X, y, MeanZero(), SEArd(zeros(d), 0.0), -1.0)
        pk = GPE(X, y, MeanZero(), SEArd(zeros(d), 0.0), -1.0, HIPCovariance(packed=true, block=256, stripe_blocks=2))
        @test pk.mll ≈ cpu.mll rtol = RTOL
        @test predict_f(pk, Xtest)[2] ≈ predict_f(cpu, Xtest)[2] rtol = 1e-5 atol = 1e-9
        update_target_and_dtarget!(cpu); update_target_and_dtarget!(pk)
        @test pk.dtarget ≈ cpu.dtarget rtol = 1e-6
    end

    @testset "FITC vs the reference's FITC(...) (src/sparse/fully_indep_train_conditional.jl)" begin
        Xu = X[:, 1:7:end]
        k = SEArd(zeros(d), 0.0) + Mat52Iso(log(0.7), log(0.5))                   # the Matern term keeps Kuu well conditioned
        cpu = GaussianProcesses.FITC(X, Xu, y, MeanConst(0.1), deepcopy(k), log(0.1))
        hip = FITC_hip(X, Xu, y, MeanConst(0.1), deepcopy(k), log(0.1))
        @test hip.mll ≈ cpu.mll rtol = 1e-6
        μc, σc = predict_f(cpu, Xtest); μh, σh = predict_f(hip, Xtest)
        @test μh ≈ μc rtol = 1e-5
        @test σh ≈ σc rtol = 1e-5 atol = 1e-8
        @test GaussianProcesses.predict_full(hip, Xtest)[2] ≈ predict_f(cpu, Xtest; full_cov=true)[2] rtol = 1e-5 atol = 1e-8
        @test update_cK!(hip) === hip.cK
        update_target_and_dtarget!(cpu); update_target_and_dtarget!(hip)
        @test hip.dtarget ≈ cpu.dtarget rtol = 1e-5 atol = 1e-7 * maximum(abs, cpu.dtarget)
    end

    @testset "the reference's printed transcript (docs/src/Regression.md:28-63,83-89)" begin
        # the document seeds the GLOBAL RNG of a Julia in which that was a MersenneTwister; an explicit one reproduces its stream on any version
        rng = MersenneTwister(20140430)
        nn = 10; x = 2π * rand(rng, nn); yy = sin.(x) + 0.05 * randn(rng, nn)
        @test x[1] ≈ 4.85461 atol = 6e-6                                           # Regression.md:58 — the stream is the document's
        gp = GPE(x', yy, MeanZero(), SE(0.0, 0.0), -1.0, HIPCovariance())
        @test gp.mll ≈ -6.335 atol = 5e-4                                          # Regression.md:63
        μ, σ² = predict_y(gp, collect(range(0, stop=2π, length=100)))
        @test μ[1:3] ≈ [0.357625, 0.384852, 0.412943] atol = 6e-7                  # Regression.md:83-89
        @test σ²[1:3] ≈ [0.603651, 0.557693, 0.512299] atol = 6e-7
    end
end
