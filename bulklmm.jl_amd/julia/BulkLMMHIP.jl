# BulkLMMHIP.jl -- thin `ccall` host for libbulklmm_hip.so (include/bulklmm_hip.h).
#
# Same signatures, keyword defaults and NamedTuple fields as BulkLMM.jl's bulkscan hot path
# (src/bulkscan.jl:81-162,188-314,321-397,428-526; src/scan.jl:94-271,485-557; src/kinship.jl:4-14), so that
#     using BulkLMMHIP: bulkscan, bulkscan_null, bulkscan_null_grid, bulkscan_alt_grid, scan, calcKinship
# is a drop-in for `using BulkLMM` on that path.  NOTE: there is no Julia in the build container, so THIS FILE HAS NEVER
# BEEN EXECUTED; it is the binding a maintainer adds (INTEGRATION.md) and mirrors bulklmm.jl_amd/api.py (the ctypes host
# that IS tested) 1:1.  What can be checked without Julia is checked as TEXT by tests/test_binding_kwargs.py: every keyword of
# the reference's signatures is accepted here and in api.py; the fields of BlmmOpts / BlmmStatus / BlmmMultiOpts (names, types,
# order, and through them offsets and sizes, against `offsetof` / `sizeof` printed by a C program compiled from the header);
# every `ccall`'s symbol, return type and argument tuple (arity and types) against the prototype in include/bulklmm_hip.h.
module BulkLMMHIP

export bulkscan_reduced, DeviceLOD, lod_columns, set_tuning, calcKinship, bulkscan, bulkscan_null, bulkscan_null_grid, bulkscan_alt_grid, bulkscan_alt_exact, scan, bulkscan_multi, lod2log10p, get_thresholds,
       lod_threshold, lod_colmax, pinned_matrix, host_register, host_unregister

const libblmm = get(ENV, "BULKLMM_HIP_LIB", joinpath(@__DIR__, "..", "csrc", "libbulklmm_hip.so"))

struct BlmmOpts            # include/bulklmm_hip.h: blmm_opts
    method::Int32; reml::Int32; add_intercept::Int32; decomp_scheme::Int32
    optim_interval::Int32; compat_flags::Int32
    prior_variance::Float64; prior_sample_size::Float64
end

mutable struct BlmmStatus  # include/bulklmm_hip.h: blmm_status
    n_neg_eig::Int64; n_nonpos_weight::Int64; n_zero_norm::Int64; n_nan_lod::Int64
    n_brent_maxiter::Int64; jacobi_sweeps::Int64; jacobi_cycles::Int64; jacobi_ticks_100mhz::Int64
    lowrank_rank::Int64; lowrank_fallback::Int64; lowrank_shared::Int64; lowrank_resid::Float64
    t_eigen_ms::Float64; t_rotate_ms::Float64; t_h2_ms::Float64; t_prep_ms::Float64; t_scan_ms::Float64; t_total_ms::Float64
    n_h2_boundary::Int64; n_h2_multimodal::Int64; n_illcond_rescan::Int64
    BlmmStatus() = new(0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0, 0, 0)
end

const NULL_EXACT, NULL_GRID, ALT_GRID = Int32(0), Int32(1), Int32(2)
const _ctx = Ref{Ptr{Cvoid}}(C_NULL)

function context(device::Integer = 0)
    if _ctx[] == C_NULL
        h = Ref{Ptr{Cvoid}}(C_NULL)
        rc = ccall((:blmm_create, libblmm), Cint, (Cint, Ptr{Cvoid}, Ref{Ptr{Cvoid}}), device, C_NULL, h)
        rc == 0 || error(unsafe_string(ccall((:blmm_err_string, libblmm), Cstring, (Cint,), rc)))
        _ctx[] = h[]
    end
    return _ctx[]
end

check(rc) = rc == 0 || error(unsafe_string(ccall((:blmm_last_error, libblmm), Cstring, (Ptr{Cvoid},), context())))

function raise_status(st)   # BlmmStatus, or one element of the per-device Vector of bulkscan_multi
    st.n_neg_eig > 0 && @warn "Negative eigenvalues exist. The kinship matrix supplied may not be SPD."   # src/transform_helpers.jl:29
    st.n_nonpos_weight > 0 && @warn "Some weights are not positive."                                      # src/wls.jl:36
    st.n_zero_norm > 0 && error("Dividing by zeros: the input vector can not contain any zeros!")         # src/util.jl:70
end

decomp(s::String) = s == "eigen" ? Int32(0) : s == "svd" ? Int32(1) : Int32(99)
# The one capability gap against the reference (LAPACK's eigen has no size limit, src/transform_helpers.jl:21-34): the device
# eigensolver stops at 2048 individuals.  Refused here, before anything is uploaded, with the library's own message.
const MAX_INDIVIDUALS = 2048
check_n(n) = n <= MAX_INDIVIDUALS || error("more than 2048 individuals: the device eigensolver (tridiagonalisation + divide and conquer) stops at n = 2048")
ptr_or_null(x) = x === missing || x === nothing ? Ptr{Float64}(C_NULL) : pointer(x)

function calcKinship(geno::Array{Float64, 2})
    (n, p) = size(geno)
    K = Array{Float64, 2}(undef, n, n)
    GC.@preserve geno K check(ccall((:blmm_kinship, libblmm), Cint, (Ptr{Cvoid}, Ptr{Float64}, Int64, Int64, Ptr{Float64}),
                                    context(), geno, n, p, K))
    return K
end

function _bulkscan(method::Int32, Y::Array{Float64, 2}, G::Array{Float64, 2}, Covar, K::Array{Float64, 2}, grid::Vector{Float64};
                   addIntercept::Bool, weights, prior_variance::Float64, prior_sample_size::Float64, reml::Bool,
                   optim_interval::Int64, decomp_scheme::String, keep_on_device::Bool = false, pvals_df::Int64 = 0)
    (n, m) = size(Y); p = size(G, 2)
    (size(G, 1) != n || size(K, 1) != n || size(K, 2) != n) && error("Dimension mismatch.")
    check_n(n)   # src/transform_helpers.jl:9-11
    (Covar !== nothing && size(Covar, 1) != n) && error("Dimension mismatch.")
    (weights !== missing && length(weights) != n) && error("Dimension mismatch.")
    ncov = Covar === nothing ? 0 : size(Covar, 2)
    o = BlmmOpts(method, reml, Covar === nothing ? true : addIntercept, decomp(decomp_scheme), optim_interval, 0,
                 prior_variance, prior_sample_size)
    # keep_on_device: L_out == NULL -- the matrix stays in the context's HBM workspace and `L` is a DeviceLOD handle
    L = keep_on_device ? nothing : Array{Float64, 2}(undef, p, m)
    h2 = method == ALT_GRID ? Array{Float64, 2}(undef, p, m) : Array{Float64, 1}(undef, m)
    st = BlmmStatus()
    # `output_pvals`: asked for right in front of the scan (after every check above), so that the scan kernels write -log10 p from
    # their epilogues (chisq_df = 1, null-* methods; otherwise the column pass runs inside the call) into a buffer of the context
    # that _last_log10p hands out; the library consumes the request first thing in the call, whatever happens next
    if pvals_df > 0
        check(ccall((:blmm_set_log10p_output, libblmm), Cint, (Ptr{Cvoid}, Ptr{Float64}, Int64, Int64),
                    context(), Ptr{Float64}(C_NULL), Int64(0), pvals_df))
    end
    GC.@preserve Y G Covar K weights grid L h2 begin
        check(ccall((:blmm_bulkscan, libblmm), Cint,
                    (Ptr{Cvoid}, Ref{BlmmOpts}, Ptr{Float64}, Int64, Int64, Ptr{Float64}, Int64, Ptr{Float64}, Int64,
                     Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Int64, Ptr{Float64}, Ptr{Float64}, Ref{BlmmStatus}),
                    context(), o, Y, n, m, G, p, ptr_or_null(Covar), ncov, K, ptr_or_null(weights), grid, length(grid),
                    ptr_or_null(L), h2, st))
    end
    raise_status(st)
    return (keep_on_device ? DeviceLOD(p, m) : L), h2
end

# ---- the LOD matrix of a `keep_on_device = true` call: it stays in HBM (2.08 GB at BXD size: 36 of a call's 39 ms are its trip over
# PCIe) and is reduced there -- what README.md:246-255, 354-359 and get_thresholds do with L -- until the context's next call
struct DeviceLOD
    p::Int64
    m::Int64
end
Base.size(d::DeviceLOD) = (d.p, d.m)
function _alive(d::DeviceLOD)
    pp = Ref{Int64}(0); mm = Ref{Int64}(0)
    rc = ccall((:blmm_last_dims, libblmm), Cint, (Ptr{Cvoid}, Ref{Int64}, Ref{Int64}), context(), pp, mm)
    (rc == 0 && pp[] == d.p && mm[] == d.m) || error("the device-resident LOD matrix has been replaced by a later call")
end
function lod_colmax(d::DeviceLOD)
    _alive(d)
    mx = Vector{Float64}(undef, d.m); arg = Vector{Int64}(undef, d.m)
    GC.@preserve mx arg check(ccall((:blmm_last_lod_colmax, libblmm), Cint, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Int64}), context(), mx, arg))
    return (max = mx, argmax = arg .+ 1)
end
function lod_threshold(d::DeviceLOD, thr::Float64; cap::Int64 = 65536)
    _alive(d)
    while true
        ii = Vector{Int32}(undef, cap); jj = Vector{Int32}(undef, cap); ll = Vector{Float64}(undef, cap)
        cnt = Ref{Int64}(0)
        GC.@preserve ii jj ll check(ccall((:blmm_last_lod_threshold, libblmm), Cint,
            (Ptr{Cvoid}, Float64, Int64, Ptr{Int32}, Ptr{Int32}, Ptr{Float64}, Ref{Int64}), context(), thr, cap, ii, jj, ll, cnt))
        if cnt[] <= cap
            k = cnt[]
            ord = sortperm(collect(zip(jj[1:k], ii[1:k])))
            return (marker = Int.(ii[1:k][ord]) .+ 1, trait = Int.(jj[1:k][ord]) .+ 1, lod = ll[1:k][ord])
        end
        cap = cnt[]
    end
end
function get_thresholds(d::DeviceLOD, signif_level::Array{Float64, 1})
    _alive(d)
    probs = 1.0 .- signif_level
    thrs = similar(probs)
    GC.@preserve probs thrs check(ccall((:blmm_last_get_thresholds, libblmm), Cint, (Ptr{Cvoid}, Ptr{Float64}, Int64, Ptr{Float64}),
                                        context(), probs, length(probs), thrs))
    return (probs = probs, thrs = thrs)
end
# L[:, traits] (1-based): the LOD profiles of a few traits -- the only part of L that crosses PCIe
function lod_columns(d::DeviceLOD, traits::Vector{Int64})
    _alive(d)
    cols = traits .- 1
    out = Array{Float64, 2}(undef, d.p, length(cols))
    GC.@preserve cols out check(ccall((:blmm_last_lod_columns, libblmm), Cint, (Ptr{Cvoid}, Ptr{Int64}, Int64, Ptr{Float64}),
                                      context(), cols, length(cols), out))
    return out
end
Base.Array(d::DeviceLOD) = lod_columns(d, collect(1:d.m))

# ---- tuning (include/bulklmm_hip.h): the switches that select another arithmetic path are properties of the context
set_tuning(key::String, value::Real) = check(ccall((:blmm_set_tuning, libblmm), Cint, (Ptr{Cvoid}, Cstring, Float64), context(), key, Float64(value)))

# ---- bulkscan WITHOUT the LOD matrix (blmm_bulkscan_reduced): per trait the peak LOD and its marker and, `threshold` given, every
# (marker, trait, LOD) with LOD > threshold, out of the scan kernels' epilogues; L is never written.  Not in the reference, whose
# users reduce L on the CPU (README.md:246-255, 354-359).
struct BlmmReduced   # include/bulklmm_hip.h: blmm_reduced
    colmax::Ptr{Float64}; argmax::Ptr{Int64}; want_triplets::Int64; thr::Float64; cap::Int64
    ti::Ptr{Int32}; tj::Ptr{Int32}; tlod::Ptr{Float64}; count::Ptr{Int64}
end
function bulkscan_reduced(Y::Array{Float64, 2}, G::Array{Float64, 2}, Covar::Union{Nothing, Array{Float64, 2}}, K::Array{Float64, 2};
                          method::String = "null-grid", h2_grid::Array{Float64, 1} = collect(0.0:0.1:0.9),
                          threshold::Union{Nothing, Float64} = nothing, cap::Int64 = 1048576, addIntercept::Bool = true,
                          weights::Union{Missing, Array{Float64, 1}} = missing, prior_variance::Float64 = 1.0,
                          prior_sample_size::Float64 = 0.0, reml::Bool = false, optim_interval::Int64 = 1,
                          decomp_scheme::String = "eigen")
    meth = method == "null-exact" ? NULL_EXACT : method == "null-grid" ? NULL_GRID : method == "alt-grid" ? ALT_GRID :
           error("Unknown method `$method`; choose null-exact, null-grid or alt-grid.")
    (n, m) = size(Y); p = size(G, 2)
    (size(G, 1) != n || size(K, 1) != n || size(K, 2) != n) && error("Dimension mismatch.")
    check_n(n)
    (Covar !== nothing && size(Covar, 1) != n) && error("Dimension mismatch.")
    (weights !== missing && length(weights) != n) && error("Dimension mismatch.")
    ncov = Covar === nothing ? 0 : size(Covar, 2)
    o = BlmmOpts(meth, reml, Covar === nothing ? true : addIntercept, decomp(decomp_scheme), optim_interval, 0,
                 prior_variance, prior_sample_size)
    grid = meth == NULL_EXACT ? Float64[] : h2_grid
    mx = Vector{Float64}(undef, m); arg = Vector{Int64}(undef, m); h2 = Vector{Float64}(undef, m)
    want = threshold !== nothing
    st = BlmmStatus()
    while true
        c1 = max(cap, 1)
        ii = Vector{Int32}(undef, c1); jj = Vector{Int32}(undef, c1); ll = Vector{Float64}(undef, c1); cnt = zeros(Int64, 1)
        GC.@preserve Y G Covar K weights grid mx arg h2 ii jj ll cnt begin
            r = BlmmReduced(pointer(mx), pointer(arg), want ? 1 : 0, want ? threshold : 0.0, want ? cap : 0,
                            pointer(ii), pointer(jj), pointer(ll), pointer(cnt))
            check(ccall((:blmm_bulkscan_reduced, libblmm), Cint,
                        (Ptr{Cvoid}, Ref{BlmmOpts}, Ptr{Float64}, Int64, Int64, Ptr{Float64}, Int64, Ptr{Float64}, Int64,
                         Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Int64, Ref{BlmmReduced}, Ptr{Float64}, Ref{BlmmStatus}),
                        context(), o, Y, n, m, G, p, ptr_or_null(Covar), ncov, K, ptr_or_null(weights), grid, length(grid), r, h2, st))
        end
        if !want || cnt[1] <= cap
            raise_status(st)
            res = (max_lod = mx, argmax = arg .+ 1, h2_null_list = h2)
            want || return res
            k = cnt[1]
            ord = sortperm(collect(zip(jj[1:k], ii[1:k])))
            return merge(res, (marker = Int.(ii[1:k][ord]) .+ 1, trait = Int.(jj[1:k][ord]) .+ 1, lod = ll[1:k][ord]))
        end
        cap = cnt[1]
    end
end
bulkscan_reduced(Y::Array{Float64, 2}, G::Array{Float64, 2}, K::Array{Float64, 2}; kwargs...) = bulkscan_reduced(Y, G, nothing, K; kwargs...)

# nb / nt_blas are accepted and ignored (thread blocking knobs of the CPU implementation)
function bulkscan_null(Y::Array{Float64, 2}, G::Array{Float64, 2}, K::Array{Float64, 2};
                       nb::Int64 = Threads.nthreads(), nt_blas::Int64 = 1, weights = missing,
                       prior_variance::Float64 = 1.0, prior_sample_size::Float64 = 0.0, reml::Bool = false,
                       optim_interval::Int64 = 1, decomp_scheme::String = "eigen", keep_on_device::Bool = false, pvals_df::Int64 = 0)
    (L, h2) = _bulkscan(NULL_EXACT, Y, G, nothing, K, Float64[]; addIntercept = true, weights = weights,
                        prior_variance = prior_variance, prior_sample_size = prior_sample_size, reml = reml,
                        optim_interval = optim_interval, decomp_scheme = decomp_scheme, keep_on_device = keep_on_device, pvals_df = pvals_df)
    return (L = L, h2_null_list = h2)
end
function bulkscan_null(Y::Array{Float64, 2}, G::Array{Float64, 2}, Covar::Array{Float64, 2}, K::Array{Float64, 2};
                       nb::Int64 = Threads.nthreads(), nt_blas::Int64 = 1, addIntercept::Bool = true, weights = missing,
                       prior_variance::Float64 = 1.0, prior_sample_size::Float64 = 0.0, reml::Bool = false,
                       optim_interval::Int64 = 1, decomp_scheme::String = "eigen", keep_on_device::Bool = false, pvals_df::Int64 = 0)
    (L, h2) = _bulkscan(NULL_EXACT, Y, G, Covar, K, Float64[]; addIntercept = addIntercept, weights = weights,
                        prior_variance = prior_variance, prior_sample_size = prior_sample_size, reml = reml,
                        optim_interval = optim_interval, decomp_scheme = decomp_scheme, keep_on_device = keep_on_device, pvals_df = pvals_df)
    return (L = L, h2_null_list = h2)
end

function bulkscan_null_grid(Y::Array{Float64, 2}, G::Array{Float64, 2}, K::Array{Float64, 2}, grid_list::Array{Float64, 1};
                            weights = missing, prior_variance::Float64 = 1.0, prior_sample_size::Float64 = 0.0,
                            reml::Bool = false, decomp_scheme::String = "eigen", keep_on_device::Bool = false, pvals_df::Int64 = 0)
    (L, h2) = _bulkscan(NULL_GRID, Y, G, nothing, K, grid_list; addIntercept = true, weights = weights,
                        prior_variance = prior_variance, prior_sample_size = prior_sample_size, reml = reml,
                        optim_interval = 1, decomp_scheme = decomp_scheme, keep_on_device = keep_on_device, pvals_df = pvals_df)
    return (L = L, h2_null_list = h2)
end
function bulkscan_null_grid(Y::Array{Float64, 2}, G::Array{Float64, 2}, Covar::Array{Float64, 2}, K::Array{Float64, 2},
                            grid_list::Array{Float64, 1}; weights = missing, addIntercept::Bool = true,
                            prior_variance::Float64 = 1.0, prior_sample_size::Float64 = 0.0, reml::Bool = false,
                            decomp_scheme::String = "eigen", keep_on_device::Bool = false, pvals_df::Int64 = 0)
    (L, h2) = _bulkscan(NULL_GRID, Y, G, Covar, K, grid_list; addIntercept = addIntercept, weights = weights,
                        prior_variance = prior_variance, prior_sample_size = prior_sample_size, reml = reml,
                        optim_interval = 1, decomp_scheme = decomp_scheme, keep_on_device = keep_on_device, pvals_df = pvals_df)
    return (L = L, h2_null_list = h2)
end

function bulkscan_alt_grid(Y::Array{Float64, 2}, G::Array{Float64, 2}, K::Array{Float64, 2}, hsq_list::Array{Float64, 1};
                           reml::Bool = false, prior_variance::Float64 = 1.0, prior_sample_size::Float64 = 0.0,
                           weights = missing, decomp_scheme::String = "eigen", keep_on_device::Bool = false, pvals_df::Int64 = 0)
    (L, h2) = _bulkscan(ALT_GRID, Y, G, nothing, K, hsq_list; addIntercept = true, weights = weights,
                        prior_variance = prior_variance, prior_sample_size = prior_sample_size, reml = reml,
                        optim_interval = 1, decomp_scheme = decomp_scheme, keep_on_device = keep_on_device, pvals_df = pvals_df)
    return (L = L, h2_panel = h2)
end
function bulkscan_alt_grid(Y::Array{Float64, 2}, G::Array{Float64, 2}, Covar::Array{Float64, 2}, K::Array{Float64, 2},
                           hsq_list::Array{Float64, 1}; reml::Bool = false, prior_variance::Float64 = 1.0,
                           prior_sample_size::Float64 = 0.0, weights = missing, addIntercept::Bool = true,
                           decomp_scheme::String = "eigen", keep_on_device::Bool = false, pvals_df::Int64 = 0)
    (L, h2) = _bulkscan(ALT_GRID, Y, G, Covar, K, hsq_list; addIntercept = addIntercept, weights = weights,
                        prior_variance = prior_variance, prior_sample_size = prior_sample_size, reml = reml,
                        optim_interval = 1, decomp_scheme = decomp_scheme, keep_on_device = keep_on_device, pvals_df = pvals_df)
    return (L = L, h2_panel = h2)
end

# lod2log10p.(L, chisq_df) of the LOD matrix the last host-pointer call left in HBM (src/util.jl:199-206, on the GPU)
function _last_log10p(dims, chisq_df::Int64)
    P = Array{Float64}(undef, dims...)
    GC.@preserve P check(ccall((:blmm_last_log10p, libblmm), Cint, (Ptr{Cvoid}, Int64, Ptr{Float64}), context(), chisq_df, P))
    return P
end

# What the last null-exact call executed in its low-rank weights form (diagnostic; include/bulklmm_hip.h: blmm_lowrank_profile):
# (traits of the shared-weights class, [(traits, rank) per segment of the heritability axis])
function lowrank_profile()
    out = zeros(Int64, 18)
    GC.@preserve out check(ccall((:blmm_lowrank_profile, libblmm), Cint, (Ptr{Cvoid}, Ptr{Int64}), context(), out))
    return (shared = out[2], segments = [(traits = out[3 + 2s], rank = out[4 + 2s]) for s in 0:(out[1] - 1)])
end

function lod2log10p(lod::Array{Float64}, df::Int64)
    P = similar(lod)
    (p, m) = ndims(lod) == 2 ? size(lod) : (length(lod), 1)
    GC.@preserve lod P check(ccall((:blmm_lod2log10p, libblmm), Cint, (Ptr{Cvoid}, Ptr{Float64}, Int64, Int64, Int64, Ptr{Float64}),
                                   context(), lod, p, m, df, P))
    return P
end
lod2log10p(lod::Float64, df::Int64) = lod2log10p([lod], df)[1]

# bulkscan(Y, G, K; ...) and bulkscan(Y, G, Covar, K; ...): src/bulkscan.jl:81-111, 113-162
function bulkscan(Y::Array{Float64, 2}, G::Array{Float64, 2}, K::Array{Float64, 2};
                  method::String = "null-grid", h2_grid::Array{Float64, 1} = collect(0.0:0.1:0.9),
                  nb::Int64 = Threads.nthreads(), nt_blas::Int64 = 1,
                  weights::Union{Missing, Array{Float64, 1}} = missing,
                  prior_variance::Float64 = 1.0, prior_sample_size::Float64 = 0.0,
                  reml::Bool = false, optim_interval::Int64 = 1,
                  decomp_scheme::String = "eigen",
                  output_pvals::Bool = false, chisq_df::Int64 = 1, keep_on_device::Bool = false)
    # when no covariates are added, make the intercept the only covariate (src/bulkscan.jl:97-108)
    return bulkscan(Y, G, ones(size(Y, 1), 1), K; method = method, h2_grid = h2_grid, nb = nb, nt_blas = nt_blas,
                    addIntercept = false, weights = weights, prior_variance = prior_variance,
                    prior_sample_size = prior_sample_size, reml = reml, optim_interval = optim_interval,
                    decomp_scheme = decomp_scheme, output_pvals = output_pvals, chisq_df = chisq_df, keep_on_device = keep_on_device)
end
function bulkscan(Y::Array{Float64, 2}, G::Array{Float64, 2}, Covar::Array{Float64, 2}, K::Array{Float64, 2};
                  method::String = "null-grid", h2_grid::Array{Float64, 1} = collect(0.0:0.1:0.9),
                  nb::Int64 = Threads.nthreads(), nt_blas::Int64 = 1, addIntercept::Bool = true,
                  weights::Union{Missing, Array{Float64, 1}} = missing,
                  prior_variance::Float64 = 1.0, prior_sample_size::Float64 = 0.0,
                  reml::Bool = false, optim_interval::Int64 = 1,
                  decomp_scheme::String = "eigen",
                  output_pvals::Bool = false, chisq_df::Int64 = 1, keep_on_device::Bool = false)
    # keep_on_device = true (not in the reference): `L` is a DeviceLOD -- the p x m matrix stays in HBM and lod_colmax /
    # lod_threshold / get_thresholds / lod_columns reduce it there
    pv = output_pvals ? chisq_df : Int64(0)
    if method == "null-exact"
        res = bulkscan_null(Y, G, Covar, K; addIntercept = addIntercept, weights = weights, prior_variance = prior_variance,
                            prior_sample_size = prior_sample_size, reml = reml, optim_interval = optim_interval,
                            decomp_scheme = decomp_scheme, keep_on_device = keep_on_device, pvals_df = pv)
    elseif method == "null-grid"
        res = bulkscan_null_grid(Y, G, Covar, K, h2_grid; addIntercept = addIntercept, weights = weights,
                                 prior_variance = prior_variance, prior_sample_size = prior_sample_size, reml = reml,
                                 decomp_scheme = decomp_scheme, keep_on_device = keep_on_device, pvals_df = pv)
    elseif method == "alt-grid"
        res = bulkscan_alt_grid(Y, G, Covar, K, h2_grid; addIntercept = addIntercept, weights = weights,
                                prior_variance = prior_variance, prior_sample_size = prior_sample_size, reml = reml,
                                decomp_scheme = decomp_scheme, keep_on_device = keep_on_device, pvals_df = pv)
    else
        error("Unknown method `$method`; choose null-exact, null-grid or alt-grid.")  # the reference hits an UndefVarError here
    end
    if output_pvals   # src/bulkscan.jl:154-157
        return merge(res, (log10Pvals_mat = _last_log10p(size(res.L), chisq_df), Chisq_df = chisq_df))
    end
    return res
end

# ---- the bulk form of scan(...; assumption = "alt") (scan_alt, src/scan.jl:397-453; not in the reference, which has the
# single-trait function and the grid approximation bulkscan_alt_grid): the exact heritability of EVERY (trait, marker) test
function bulkscan_alt_exact(Y::Array{Float64, 2}, G::Array{Float64, 2}, Covar::Array{Float64, 2}, K::Array{Float64, 2};
                            addIntercept::Bool = true, weights::Union{Missing, Array{Float64, 1}} = missing,
                            prior_variance::Float64 = 0.0, prior_sample_size::Float64 = 0.0, reml::Bool = false,
                            optim_interval::Int64 = 1, decomp_scheme::String = "eigen")
    n = size(Y, 1); m = size(Y, 2); p = size(G, 2)
    (size(G, 1) != n || size(K, 1) != n || size(K, 2) != n || size(Covar, 1) != n) && error("Dimension mismatch.")
    check_n(n)
    (weights !== missing && length(weights) != n) && error("Dimension mismatch.")
    o = BlmmOpts(NULL_EXACT, reml, addIntercept, decomp(decomp_scheme), optim_interval, 0, prior_variance, prior_sample_size)
    L = Array{Float64, 2}(undef, p, m); H = Array{Float64, 2}(undef, p, m)
    h2 = Array{Float64, 1}(undef, m); s2 = Array{Float64, 1}(undef, m); st = BlmmStatus()
    GC.@preserve Y G Covar K weights L H h2 s2 check(ccall((:blmm_bulkscan_alt_exact, libblmm), Cint,
        (Ptr{Cvoid}, Ref{BlmmOpts}, Ptr{Float64}, Int64, Int64, Ptr{Float64}, Int64, Ptr{Float64}, Int64, Ptr{Float64}, Ptr{Float64},
         Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ref{BlmmStatus}),
        context(), o, Y, n, m, G, p, Covar, size(Covar, 2), K, ptr_or_null(weights), L, H, h2, s2, st))
    raise_status(st)
    return (L = L, h2_panel = H, h2_null_list = h2, sigma2_e = s2)
end
bulkscan_alt_exact(Y::Array{Float64, 2}, G::Array{Float64, 2}, K::Array{Float64, 2}; kwargs...) =
    bulkscan_alt_exact(Y, G, ones(size(Y, 1), 1), K; addIntercept = false, kwargs...)

# ---- several GPUs of one node in ONE call (blmm_bulkscan_multi): the trait blocks the reference deals to its threads
# (src/bulkscan.jl:263-309) go to the devices; gather = :host_shards (default) | :none | :allgather
const _mctx = Ref{Ptr{Cvoid}}(C_NULL)
const _mctx_devices = Ref{Vector{Int32}}(Int32[])
function multi_context(devices::Vector{Int32} = Int32[])
    if _mctx[] != C_NULL && _mctx_devices[] != devices      # a different device list: a new context (the old one is released)
        ccall((:blmm_destroy_multi, libblmm), Cvoid, (Ptr{Cvoid},), _mctx[])
        _mctx[] = C_NULL
    end
    if _mctx[] == C_NULL
        h = Ref{Ptr{Cvoid}}(C_NULL)
        rc = GC.@preserve devices ccall((:blmm_create_multi, libblmm), Cint, (Ptr{Int32}, Cint, Ref{Ptr{Cvoid}}),
                                        isempty(devices) ? Ptr{Int32}(C_NULL) : pointer(devices), length(devices), h)
        rc == 0 || error(unsafe_string(ccall((:blmm_err_string, libblmm), Cstring, (Cint,), rc)))
        _mctx[] = h[]
        _mctx_devices[] = copy(devices)
    end
    return _mctx[]
end
struct BlmmMultiOpts; gather_mode::Int32; reserved::Int32; end
function bulkscan_multi(Y::Array{Float64, 2}, G::Array{Float64, 2}, K::Array{Float64, 2};
                        method::String = "null-grid", h2_grid::Array{Float64, 1} = collect(0.0:0.1:0.9),
                        gather::Symbol = :host_shards, devices::Vector{Int32} = Int32[],
                        weights::Union{Missing, Array{Float64, 1}} = missing,
                        prior_variance::Float64 = 1.0, prior_sample_size::Float64 = 0.0,
                        reml::Bool = false, optim_interval::Int64 = 1, decomp_scheme::String = "eigen")
    meth = method == "null-exact" ? NULL_EXACT : method == "null-grid" ? NULL_GRID : method == "alt-grid" ? ALT_GRID :
           error("Unknown method `$method`; choose null-exact, null-grid or alt-grid.")
    (n, m) = size(Y); p = size(G, 2)
    (size(G, 1) != n || size(K, 1) != n || size(K, 2) != n) && error("Dimension mismatch.")
    check_n(n)
    (weights !== missing && length(weights) != n) && error("Dimension mismatch.")
    o = BlmmOpts(meth, reml, true, decomp(decomp_scheme), optim_interval, 0, prior_variance, prior_sample_size)
    mo = BlmmMultiOpts(gather == :none ? 0 : gather == :allgather ? 2 : 1, 0)
    L = Array{Float64, 2}(undef, p, m)
    h2 = meth == ALT_GRID ? Array{Float64, 2}(undef, p, m) : Array{Float64, 1}(undef, m)
    mc = multi_context(devices)
    ndev = ccall((:blmm_multi_ndev, libblmm), Cint, (Ptr{Cvoid},), mc)
    sts = [BlmmStatus() for _ in 1:ndev]                 # one status per device: the warnings / errors of every shard are raised
    stbuf = Vector{UInt8}(undef, ndev * sizeof(BlmmStatus))
    GC.@preserve Y G K weights h2_grid L h2 stbuf begin
        rc = ccall((:blmm_bulkscan_multi, libblmm), Cint,
                   (Ptr{Cvoid}, Ref{BlmmOpts}, Ref{BlmmMultiOpts}, Ptr{Float64}, Int64, Int64, Ptr{Float64}, Int64, Ptr{Float64}, Int64,
                    Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Int64, Ptr{Float64}, Ptr{Float64}, Ptr{Cvoid}),
                   mc, o, mo, Y, n, m, G, p, C_NULL, 0, K, ptr_or_null(weights), h2_grid, length(h2_grid), L, h2, pointer(stbuf))
        rc == 0 || error(unsafe_string(ccall((:blmm_multi_last_error, libblmm), Cstring, (Ptr{Cvoid},), mc)))
        for r in 1:ndev
            sts[r] = unsafe_load(Ptr{BlmmStatus}(pointer(stbuf)) , r)
        end
    end
    foreach(raise_status, sts)
    return meth == ALT_GRID ? (L = L, h2_panel = h2) : (L = L, h2_null_list = h2)
end

# ---- scan: the reference's four methods (src/scan.jl:94-120, 122-148, 150-180, 182-271).  On the GPU path:
#   assumption = "null" and "alt" (scan_alt, src/scan.jl:397-453: one Brent search per marker on the device, blmm_scan_alt);
#   method ("qr" / "cholesky") selects a CPU factorisation and has no meaning here: accepted, ignored;
#   profileLL = true (profile_LL of src/analysis_helpers) is not part of this path: a clear error.
function scan(y::Array{Float64, 1}, g::Array{Float64, 2}, K::Array{Float64, 2};
              weights::Union{Missing, Array{Float64, 1}} = missing,
              prior_variance::Float64 = 0.0, prior_sample_size::Float64 = 0.0, addIntercept::Bool = true,
              reml::Bool = false, assumption::String = "null", method::String = "qr", optim_interval::Int64 = 1,
              permutation_test::Bool = false, nperms::Int64 = 1024, rndseed::Int64 = 0,
              profileLL::Bool = false, markerID::Int = 0, h2_grid::Array{Float64, 1} = Array{Float64, 1}(undef, 1),
              decomp_scheme::String = "eigen",
              output_pvals::Bool = false, chisq_df::Int64 = 1,
              perm_precision::String = "f64")   # "f32" (not in the reference): L_perms on the fp32 matrix cores, as Float32
    return scan(reshape(y, :, 1), g, K; weights = weights, addIntercept = addIntercept, prior_variance = prior_variance,
                prior_sample_size = prior_sample_size, reml = reml, assumption = assumption, method = method,
                optim_interval = optim_interval, permutation_test = permutation_test, nperms = nperms, rndseed = rndseed,
                profileLL = profileLL, markerID = markerID, h2_grid = h2_grid, decomp_scheme = decomp_scheme,
                output_pvals = output_pvals, chisq_df = chisq_df, perm_precision = perm_precision)
end
function scan(y::Array{Float64, 1}, g::Array{Float64, 2}, covar::Array{Float64, 2}, K::Array{Float64, 2};
              weights::Union{Missing, Array{Float64, 1}} = missing,
              prior_variance::Float64 = 0.0, prior_sample_size::Float64 = 0.0, addIntercept::Bool = true,
              reml::Bool = false, assumption::String = "null", method::String = "qr", optim_interval::Int64 = 1,
              permutation_test::Bool = false, nperms::Int64 = 1024, rndseed::Int64 = 0,
              profileLL::Bool = false, markerID::Int = 0, h2_grid::Array{Float64, 1} = Array{Float64, 1}(undef, 1),
              decomp_scheme::String = "eigen",
              output_pvals::Bool = false, chisq_df::Int64 = 1,
              perm_precision::String = "f64")
    return scan(reshape(y, :, 1), g, covar, K; weights = weights, addIntercept = addIntercept, prior_variance = prior_variance,
                prior_sample_size = prior_sample_size, reml = reml, assumption = assumption, method = method,
                optim_interval = optim_interval, permutation_test = permutation_test, nperms = nperms, rndseed = rndseed,
                profileLL = profileLL, markerID = markerID, h2_grid = h2_grid, decomp_scheme = decomp_scheme,
                output_pvals = output_pvals, chisq_df = chisq_df, perm_precision = perm_precision)
end
function scan(y::Array{Float64, 2}, g::Array{Float64, 2}, K::Array{Float64, 2};
              weights::Union{Missing, Array{Float64, 1}} = missing,
              prior_variance::Float64 = 0.0, prior_sample_size::Float64 = 0.0, addIntercept::Bool = true,
              reml::Bool = false, assumption::String = "null", method::String = "qr", optim_interval::Int64 = 1,
              permutation_test::Bool = false, nperms::Int64 = 1024, rndseed::Int64 = 0,
              profileLL::Bool = false, markerID::Int = 0, h2_grid::Array{Float64, 1} = Array{Float64, 1}(undef, 1),
              decomp_scheme::String = "eigen",
              output_pvals::Bool = false, chisq_df::Int64 = 1,
              perm_precision::String = "f64")
    addIntercept || error("Intercept has to be added when no other covariate is given.")   # src/scan.jl:167-169
    return scan(y, g, ones(size(y, 1), 1), K; weights = weights, addIntercept = false, prior_variance = prior_variance,
                prior_sample_size = prior_sample_size, reml = reml, assumption = assumption, method = method,
                optim_interval = optim_interval, permutation_test = permutation_test, nperms = nperms, rndseed = rndseed,
                profileLL = profileLL, markerID = markerID, h2_grid = h2_grid, decomp_scheme = decomp_scheme,
                output_pvals = output_pvals, chisq_df = chisq_df, perm_precision = perm_precision)
end
function scan(y::Array{Float64, 2}, g::Array{Float64, 2}, covar::Array{Float64, 2}, K::Array{Float64, 2};
              weights::Union{Missing, Array{Float64, 1}} = missing,
              prior_variance::Float64 = 0.0, prior_sample_size::Float64 = 0.0, addIntercept::Bool = true,
              reml::Bool = false, assumption::String = "null", method::String = "qr", optim_interval::Int64 = 1,
              permutation_test::Bool = false, nperms::Int64 = 1024, rndseed::Int64 = 0,
              profileLL::Bool = false, markerID::Int = 0, h2_grid::Array{Float64, 1} = Array{Float64, 1}(undef, 1),
              decomp_scheme::String = "eigen",
              output_pvals::Bool = false, chisq_df::Int64 = 1,
              perm_precision::String = "f64")
    assumption == "alt" && permutation_test && error("Permutation test option currently is not supported for the alternative assumption.")
    assumption in ("null", "alt") || error("Assumption keyword is not supported. Please enter null or alt.")
    profileLL && error("profileLL = true (profile_LL) is not part of the GPU path; call BulkLMM.scan for it")
    size(y, 2) == 1 || error("Can only handle one trait.")                                   # src/scan.jl:496-498
    n = size(y, 1); p = size(g, 2)
    (size(g, 1) != n || size(K, 1) != n || size(K, 2) != n || size(covar, 1) != n) && error("Dimension mismatch.")
    check_n(n)
    (weights !== missing && length(weights) != n) && error("Dimension mismatch.")
    np = permutation_test ? nperms : 0
    np < 0 && error("The required number of permutations must be a positive integer.")
    perm_precision in ("f64", "f32") || error("perm_precision must be \"f64\" or \"f32\".")
    # `weights`, the intercept and the covariates go to the library as they are: it applies W (src/scan.jl:200-221) itself
    o = BlmmOpts(NULL_EXACT, reml, addIntercept, decomp(decomp_scheme), optim_interval, 0, prior_variance, prior_sample_size)
    scal = zeros(2); lod = Array{Float64, 1}(undef, p); st = BlmmStatus()
    ncov = size(covar, 2)
    if assumption == "alt"   # scan_alt (src/scan.jl:397-453): one Brent search per marker on the device
        h2_each = Array{Float64, 1}(undef, p)
        GC.@preserve y g covar K weights scal lod h2_each check(ccall((:blmm_scan_alt, libblmm), Cint,
            (Ptr{Cvoid}, Ref{BlmmOpts}, Ptr{Float64}, Int64, Ptr{Float64}, Int64, Ptr{Float64}, Int64, Ptr{Float64}, Ptr{Float64},
             Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ref{BlmmStatus}),
            context(), o, y, n, g, p, covar, ncov, K, ptr_or_null(weights), scal, lod, h2_each, st))
        raise_status(st)
        res = (sigma2_e = scal[1], h2_null = scal[2], h2_each_marker = h2_each, lod = lod)
        return output_pvals ? merge(res, (log10pvals = lod2log10p(lod, chisq_df),)) : res
    end
    f32 = perm_precision == "f32"
    Lp = f32 ? Array{Float32, 2}(undef, p, max(np, 1)) : Array{Float64, 2}(undef, p, max(np, 1))
    GC.@preserve y g covar K weights scal lod Lp begin
        if f32
            check(ccall((:blmm_scan_perms_f32, libblmm), Cint,
                        (Ptr{Cvoid}, Ref{BlmmOpts}, Ptr{Float64}, Int64, Ptr{Float64}, Int64, Ptr{Float64}, Int64, Ptr{Float64},
                         Ptr{Float64}, Int64, UInt64, Ptr{Int32}, Ptr{Float64}, Ptr{Float64}, Ptr{Float32}, Ref{BlmmStatus}),
                        context(), o, y, n, g, p, covar, ncov, K, ptr_or_null(weights), np, UInt64(rndseed), C_NULL, scal, lod, Lp, st))
        else
            check(ccall((:blmm_scan_perms, libblmm), Cint,
                        (Ptr{Cvoid}, Ref{BlmmOpts}, Ptr{Float64}, Int64, Ptr{Float64}, Int64, Ptr{Float64}, Int64, Ptr{Float64},
                         Ptr{Float64}, Int64, UInt64, Ptr{Int32}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ref{BlmmStatus}),
                        context(), o, y, n, g, p, covar, ncov, K, ptr_or_null(weights), np, UInt64(rndseed), C_NULL, scal, lod, Lp, st))
        end
    end
    raise_status(st)
    res = permutation_test ? (sigma2_e = scal[1], h2_null = scal[2], lod = lod, L_perms = Lp[:, 1:np]) :
                             (sigma2_e = scal[1], h2_null = scal[2], lod = lod)
    if output_pvals   # src/scan.jl:353-355; with permutations the reference hits an UndefVarError (src/scan.jl:551): fixed
        res = merge(res, (log10pvals = lod2log10p(lod, chisq_df),))
    end
    return res
end

# get_thresholds (src/analysis_helpers/single_trait_analysis.jl:13-23): column maxima, sort, quantile on the GPU
function get_thresholds(L_perms::Array{Float64, 2}, signif_level::Array{Float64, 1})
    probs = 1.0 .- signif_level
    thrs = similar(probs)
    GC.@preserve L_perms probs thrs check(ccall((:blmm_get_thresholds, libblmm), Cint,
        (Ptr{Cvoid}, Ptr{Float64}, Int64, Int64, Ptr{Float64}, Int64, Ptr{Float64}),
        context(), L_perms, size(L_perms, 1), size(L_perms, 2), probs, length(probs), thrs))
    return (probs = probs, thrs = thrs)
end

# ---- on-device consumers of L (README.md:246-255, 354-359) -------------------------------------------------------------
# (marker, trait, LOD) of every LOD > thr, 1-based like findall(L .> thr): the filter behind plot_eQTL(...; threshold)
function lod_threshold(L::Array{Float64, 2}, thr::Float64; cap::Int64 = max(1024, length(L) ÷ 64))
    (p, m) = size(L)
    while true
        ii = Vector{Int32}(undef, cap); jj = Vector{Int32}(undef, cap); ll = Vector{Float64}(undef, cap)
        cnt = Ref{Int64}(0)
        GC.@preserve L ii jj ll check(ccall((:blmm_lod_threshold, libblmm), Cint,
            (Ptr{Cvoid}, Ptr{Float64}, Int64, Int64, Float64, Int64, Ptr{Int32}, Ptr{Int32}, Ptr{Float64}, Ref{Int64}),
            context(), L, p, m, thr, cap, ii, jj, ll, cnt))
        if cnt[] <= cap
            k = cnt[]
            ord = sortperm(collect(zip(jj[1:k], ii[1:k])))
            return (marker = Int.(ii[1:k][ord]) .+ 1, trait = Int.(jj[1:k][ord]) .+ 1, lod = ll[1:k][ord])
        end
        cap = cnt[]
    end
end
# per-column maximum of an LOD matrix and the (1-based) marker where it sits
function lod_colmax(L::Array{Float64, 2})
    (p, m) = size(L)
    mx = Vector{Float64}(undef, m); arg = Vector{Int64}(undef, m)
    GC.@preserve L mx arg check(ccall((:blmm_lod_colmax, libblmm), Cint,
        (Ptr{Cvoid}, Ptr{Float64}, Int64, Int64, Ptr{Float64}, Ptr{Int64}), context(), L, p, m, mx, arg))
    return (max = mx, argmax = arg .+ 1)
end

# ---- pinned host memory for the outputs: L (2.08 GB at BXD size) then crosses the PCIe link in one asynchronous copy at link rate
# (42 ms against 52 ms into pageable memory).  pinned_matrix wraps blmm_host_alloc memory (released by the finalizer);
# host_register pins the pages of an Array the caller already has.
function pinned_matrix(p::Integer, m::Integer)
    ptr = ccall((:blmm_host_alloc, libblmm), Ptr{Cvoid}, (UInt64,), UInt64(8 * p * m))
    ptr == C_NULL && error("blmm_host_alloc failed")
    A = unsafe_wrap(Array, Ptr{Float64}(ptr), (Int(p), Int(m)); own = false)
    finalizer(_ -> ccall((:blmm_host_free, libblmm), Cvoid, (Ptr{Cvoid},), ptr), A)
    return A
end
host_register(A::Array{Float64}) = (ccall((:blmm_host_register, libblmm), Cint, (Ptr{Cvoid}, UInt64), A, UInt64(sizeof(A))) == 0 || error("hipHostRegister failed"); A)
host_unregister(A::Array{Float64}) = (ccall((:blmm_host_unregister, libblmm), Cint, (Ptr{Cvoid},), A); nothing)

# Readers (src/readData.jl:41-96, 159-165): the numeric table is parsed by the library's host code and copied into a Julia array
check_io(rc) = rc == 0 || error("could not read the file: " * unsafe_string(ccall((:blmm_err_string, libblmm), Cstring, (Cint,), rc)))
function read_table(open_table::Function)
    h = Ref{Ptr{Cvoid}}(C_NULL)
    check_io(open_table(h))
    try
        A = Array{Float64, 2}(undef, ccall((:blmm_table_rows, libblmm), Int64, (Ptr{Cvoid},), h[]),
                              ccall((:blmm_table_cols, libblmm), Int64, (Ptr{Cvoid},), h[]))
        check_io(ccall((:blmm_table_copy, libblmm), Cint, (Ptr{Cvoid}, Ptr{Float64}), h[], A))
        return A
    finally
        ccall((:blmm_table_free, libblmm), Cvoid, (Ptr{Cvoid},), h[])
    end
end
read_csv(file, skip, first, step, drop) = read_table(h -> ccall((:blmm_read_csv, libblmm), Cint,
    (Cstring, Int64, Int64, Int64, Int64, Ref{Ptr{Cvoid}}), file, skip, first, step, drop, h))
readGenoProb(file::AbstractString) = read_csv(file, 1, 1, 1, 0)                       # header line and id column dropped (getmarkernames = getids = true)
readGenoProb_ExcludeComplements(file::AbstractString) = read_csv(file, 1, 1, 2, 0)
readBXDpheno(file::AbstractString) = read_csv(file, 1, 1, 1, 1)
readBXDgeno(file::AbstractString; skipstart = 1) = read_csv(file, skipstart, 1, 2, 0)
readhe(file::AbstractString) = read_table(h -> ccall((:blmm_read_he, libblmm), Cint, (Cstring, Ref{Ptr{Cvoid}}), file, h))

# round.(calcKinship(G), digits = d) in one device call (README.md:176-181)
function calcKinship(G::Array{Float64, 2}, digits::Integer)
    n, p = size(G)
    K = Array{Float64, 2}(undef, n, n)
    check(ccall((:blmm_kinship_rounded, libblmm), Cint, (Ptr{Cvoid}, Ptr{Float64}, Int64, Int64, Int64, Ptr{Float64}),
                context(), G, n, p, digits, K))
    return K
end

end # module
