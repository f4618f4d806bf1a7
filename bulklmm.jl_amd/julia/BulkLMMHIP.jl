# BulkLMMHIP.jl -- thin `ccall` host for libbulklmm_hip.so (include/bulklmm_hip.h).
#
# Same signatures, keyword defaults and NamedTuple fields as BulkLMM.jl's bulkscan hot path
# (src/bulkscan.jl:81-162,188-314,321-397,428-526; src/scan.jl:94-271,485-557; src/kinship.jl:4-14), so that
#     using BulkLMMHIP: bulkscan, bulkscan_null, bulkscan_null_grid, bulkscan_alt_grid, scan, calcKinship
# is a drop-in for `using BulkLMM` on that path.  NOTE: there is no Julia in the build container; this file is the
# binding a maintainer adds (INTEGRATION.md) and mirrors bulklmm.jl_amd/api.py (the ctypes host that IS tested) 1:1.
module BulkLMMHIP

export calcKinship, bulkscan, bulkscan_null, bulkscan_null_grid, bulkscan_alt_grid, scan

const libblmm = get(ENV, "BULKLMM_HIP_LIB", joinpath(@__DIR__, "..", "csrc", "libbulklmm_hip.so"))

struct BlmmOpts            # include/bulklmm_hip.h: blmm_opts
    method::Int32; reml::Int32; add_intercept::Int32; decomp_scheme::Int32
    optim_interval::Int32; compat_flags::Int32
    prior_variance::Float64; prior_sample_size::Float64
end

mutable struct BlmmStatus  # include/bulklmm_hip.h: blmm_status
    n_neg_eig::Int64; n_nonpos_weight::Int64; n_zero_norm::Int64; n_nan_lod::Int64
    n_brent_maxiter::Int64; jacobi_sweeps::Int64; jacobi_cycles::Int64; jacobi_ticks_100mhz::Int64
    lowrank_rank::Int64; lowrank_resid::Float64
    t_eigen_ms::Float64; t_rotate_ms::Float64; t_h2_ms::Float64; t_prep_ms::Float64; t_scan_ms::Float64; t_total_ms::Float64
    BlmmStatus() = new(0, 0, 0, 0, 0, 0, 0, 0, 0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0)
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

function raise_status(st::BlmmStatus)
    st.n_neg_eig > 0 && @warn "Negative eigenvalues exist. The kinship matrix supplied may not be SPD."   # src/transform_helpers.jl:29
    st.n_nonpos_weight > 0 && @warn "Some weights are not positive."                                      # src/wls.jl:36
    st.n_zero_norm > 0 && error("Dividing by zeros: the input vector can not contain any zeros!")         # src/util.jl:70
end

decomp(s::String) = s == "eigen" ? Int32(0) : s == "svd" ? Int32(1) : Int32(99)
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
                   optim_interval::Int64, decomp_scheme::String)
    (n, m) = size(Y); p = size(G, 2)
    (size(G, 1) != n || size(K, 1) != n) && error("Dimension mismatch.")
    ncov = Covar === nothing ? 0 : size(Covar, 2)
    o = BlmmOpts(method, reml, Covar === nothing ? true : addIntercept, decomp(decomp_scheme), optim_interval, 0,
                 prior_variance, prior_sample_size)
    L = Array{Float64, 2}(undef, p, m)
    h2 = method == ALT_GRID ? Array{Float64, 2}(undef, p, m) : Array{Float64, 1}(undef, m)
    st = BlmmStatus()
    GC.@preserve Y G Covar K weights grid L h2 begin
        check(ccall((:blmm_bulkscan, libblmm), Cint,
                    (Ptr{Cvoid}, Ref{BlmmOpts}, Ptr{Float64}, Int64, Int64, Ptr{Float64}, Int64, Ptr{Float64}, Int64,
                     Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Int64, Ptr{Float64}, Ptr{Float64}, Ref{BlmmStatus}),
                    context(), o, Y, n, m, G, p, ptr_or_null(Covar), ncov, K, ptr_or_null(weights), grid, length(grid), L, h2, st))
    end
    raise_status(st)
    return L, h2
end

# nb / nt_blas are accepted and ignored (thread blocking knobs of the CPU implementation)
function bulkscan_null(Y::Array{Float64, 2}, G::Array{Float64, 2}, K::Array{Float64, 2};
                       nb::Int64 = Threads.nthreads(), nt_blas::Int64 = 1, weights = missing,
                       prior_variance::Float64 = 1.0, prior_sample_size::Float64 = 0.0, reml::Bool = false,
                       optim_interval::Int64 = 1, decomp_scheme::String = "eigen")
    (L, h2) = _bulkscan(NULL_EXACT, Y, G, nothing, K, Float64[]; addIntercept = true, weights = weights,
                        prior_variance = prior_variance, prior_sample_size = prior_sample_size, reml = reml,
                        optim_interval = optim_interval, decomp_scheme = decomp_scheme)
    return (L = L, h2_null_list = h2)
end
function bulkscan_null(Y::Array{Float64, 2}, G::Array{Float64, 2}, Covar::Array{Float64, 2}, K::Array{Float64, 2};
                       nb::Int64 = Threads.nthreads(), nt_blas::Int64 = 1, addIntercept::Bool = true, weights = missing,
                       prior_variance::Float64 = 1.0, prior_sample_size::Float64 = 0.0, reml::Bool = false,
                       optim_interval::Int64 = 1, decomp_scheme::String = "eigen")
    (L, h2) = _bulkscan(NULL_EXACT, Y, G, Covar, K, Float64[]; addIntercept = addIntercept, weights = weights,
                        prior_variance = prior_variance, prior_sample_size = prior_sample_size, reml = reml,
                        optim_interval = optim_interval, decomp_scheme = decomp_scheme)
    return (L = L, h2_null_list = h2)
end

function bulkscan_null_grid(Y::Array{Float64, 2}, G::Array{Float64, 2}, K::Array{Float64, 2}, grid_list::Array{Float64, 1};
                            weights = missing, prior_variance::Float64 = 1.0, prior_sample_size::Float64 = 0.0,
                            reml::Bool = false, decomp_scheme::String = "eigen")
    (L, h2) = _bulkscan(NULL_GRID, Y, G, nothing, K, grid_list; addIntercept = true, weights = weights,
                        prior_variance = prior_variance, prior_sample_size = prior_sample_size, reml = reml,
                        optim_interval = 1, decomp_scheme = decomp_scheme)
    return (L = L, h2_null_list = h2)
end
function bulkscan_null_grid(Y::Array{Float64, 2}, G::Array{Float64, 2}, Covar::Array{Float64, 2}, K::Array{Float64, 2},
                            grid_list::Array{Float64, 1}; weights = missing, addIntercept::Bool = true,
                            prior_variance::Float64 = 1.0, prior_sample_size::Float64 = 0.0, reml::Bool = false,
                            decomp_scheme::String = "eigen")
    (L, h2) = _bulkscan(NULL_GRID, Y, G, Covar, K, grid_list; addIntercept = addIntercept, weights = weights,
                        prior_variance = prior_variance, prior_sample_size = prior_sample_size, reml = reml,
                        optim_interval = 1, decomp_scheme = decomp_scheme)
    return (L = L, h2_null_list = h2)
end

function bulkscan_alt_grid(Y::Array{Float64, 2}, G::Array{Float64, 2}, K::Array{Float64, 2}, hsq_list::Array{Float64, 1};
                           reml::Bool = false, prior_variance::Float64 = 1.0, prior_sample_size::Float64 = 0.0,
                           weights = missing, decomp_scheme::String = "eigen")
    (L, h2) = _bulkscan(ALT_GRID, Y, G, nothing, K, hsq_list; addIntercept = true, weights = weights,
                        prior_variance = prior_variance, prior_sample_size = prior_sample_size, reml = reml,
                        optim_interval = 1, decomp_scheme = decomp_scheme)
    return (L = L, h2_panel = h2)
end
function bulkscan_alt_grid(Y::Array{Float64, 2}, G::Array{Float64, 2}, Covar::Array{Float64, 2}, K::Array{Float64, 2},
                           hsq_list::Array{Float64, 1}; reml::Bool = false, prior_variance::Float64 = 1.0,
                           prior_sample_size::Float64 = 0.0, weights = missing, addIntercept::Bool = true,
                           decomp_scheme::String = "eigen")
    (L, h2) = _bulkscan(ALT_GRID, Y, G, Covar, K, hsq_list; addIntercept = addIntercept, weights = weights,
                        prior_variance = prior_variance, prior_sample_size = prior_sample_size, reml = reml,
                        optim_interval = 1, decomp_scheme = decomp_scheme)
    return (L = L, h2_panel = h2)
end

function bulkscan(Y::Array{Float64, 2}, G::Array{Float64, 2}, K::Array{Float64, 2};
                  method::String = "null-grid", h2_grid::Array{Float64, 1} = collect(0.0:0.1:0.9),
                  nb::Int64 = Threads.nthreads(), nt_blas::Int64 = 1, weights = missing,
                  prior_variance::Float64 = 1.0, prior_sample_size::Float64 = 0.0, reml::Bool = false,
                  optim_interval::Int64 = 1, decomp_scheme::String = "eigen")
    if method == "null-exact"
        return bulkscan_null(Y, G, K; weights = weights, prior_variance = prior_variance, prior_sample_size = prior_sample_size,
                             reml = reml, optim_interval = optim_interval, decomp_scheme = decomp_scheme)
    elseif method == "null-grid"
        return bulkscan_null_grid(Y, G, K, h2_grid; weights = weights, prior_variance = prior_variance,
                                  prior_sample_size = prior_sample_size, reml = reml, decomp_scheme = decomp_scheme)
    elseif method == "alt-grid"
        return bulkscan_alt_grid(Y, G, K, h2_grid; weights = weights, prior_variance = prior_variance,
                                 prior_sample_size = prior_sample_size, reml = reml, decomp_scheme = decomp_scheme)
    end
    error("Unknown method `$method`; choose null-exact, null-grid or alt-grid.")  # the reference hits an UndefVarError here
end

# scan(y, G, K; ...): null assumption, optional permutation test (src/scan.jl:94-271, 485-557)
function scan(y::Array{Float64, 1}, g::Array{Float64, 2}, K::Array{Float64, 2};
              weights = missing, prior_variance::Float64 = 0.0, prior_sample_size::Float64 = 0.0, addIntercept::Bool = true,
              reml::Bool = false, assumption::String = "null", optim_interval::Int64 = 1,
              permutation_test::Bool = false, nperms::Int64 = 1024, rndseed::Int64 = 0, decomp_scheme::String = "eigen",
              perm_precision::String = "f64")   # "f32": L_perms on the fp32 matrix cores, returned as Float32
    addIntercept || error("Intercept has to be added when no other covariate is given.")
    assumption == "null" || error(assumption == "alt" ? "scan_alt is not part of the GPU path" :
                                  "Assumption keyword is not supported. Please enter null or alt.")
    n = length(y); p = size(g, 2)
    np = permutation_test ? nperms : 0
    np < 0 && error("The required number of permutations must be a positive integer.")
    o = BlmmOpts(NULL_EXACT, reml, true, decomp(decomp_scheme), optim_interval, 0, prior_variance, prior_sample_size)
    perm_precision in ("f64", "f32") || error("perm_precision must be \"f64\" or \"f32\".")
    scal = zeros(2); lod = Array{Float64, 1}(undef, p); st = BlmmStatus()
    if perm_precision == "f32"
        Lp32 = Array{Float32, 2}(undef, p, max(np, 1))
        GC.@preserve y g K weights scal lod Lp32 begin
            check(ccall((:blmm_scan_perms_f32, libblmm), Cint,
                        (Ptr{Cvoid}, Ref{BlmmOpts}, Ptr{Float64}, Int64, Ptr{Float64}, Int64, Ptr{Float64}, Int64, Ptr{Float64},
                         Ptr{Float64}, Int64, UInt64, Ptr{Int32}, Ptr{Float64}, Ptr{Float64}, Ptr{Float32}, Ref{BlmmStatus}),
                        context(), o, y, n, g, p, C_NULL, 0, K, ptr_or_null(weights), np, UInt64(rndseed), C_NULL, scal, lod, Lp32, st))
        end
        raise_status(st)
        return permutation_test ? (sigma2_e = scal[1], h2_null = scal[2], lod = lod, L_perms = Lp32[:, 1:np]) :
                                  (sigma2_e = scal[1], h2_null = scal[2], lod = lod)
    end
    Lp = Array{Float64, 2}(undef, p, max(np, 1))
    GC.@preserve y g K weights scal lod Lp begin
        check(ccall((:blmm_scan_perms, libblmm), Cint,
                    (Ptr{Cvoid}, Ref{BlmmOpts}, Ptr{Float64}, Int64, Ptr{Float64}, Int64, Ptr{Float64}, Int64, Ptr{Float64},
                     Ptr{Float64}, Int64, UInt64, Ptr{Int32}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ref{BlmmStatus}),
                    context(), o, y, n, g, p, C_NULL, 0, K, ptr_or_null(weights), np, UInt64(rndseed), C_NULL, scal, lod, Lp, st))
    end
    raise_status(st)
    return permutation_test ? (sigma2_e = scal[1], h2_null = scal[2], lod = lod, L_perms = Lp[:, 1:np]) :
                              (sigma2_e = scal[1], h2_null = scal[2], lod = lod)
end

end # module
