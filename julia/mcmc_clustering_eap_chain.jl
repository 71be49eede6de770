# mcmc_clustering_eap_chain.jl -- Julia host of the MI355X path for the reference's clustering main.
#
# Same command line, same two CSV files, same twelve stdout lines as the reference's
# mcmc_clustering_eap_chain.jl; the step loop (single-monomer move + cluster_flip!) runs on the GPU
# through libpstat (C ABI, include/pstat.h, move_set = PSTAT_MOVES_CLUSTER) via ccall.
# NOT EXECUTED IN THE BUILD IMAGE (no Julia toolchain there); it is the twin of
# polymer_stats_amd/mcmc_clustering_eap_chain.py, which the test-suite exercises.  Only ArgParse is needed.
#
#   julia julia/mcmc_clustering_eap_chain.jl -n 100 -e 1 -K 1 -u Ising -N 2500000 --burn-in 100000 \
#         --num-chains 16384 --prefix out/run1 -v 2
using ArgParse
using Logging
using Random

const LIBPSTAT = get(ENV, "PSTAT_LIB", joinpath(@__DIR__, "..", "polymer_stats_amd", "libpstat.so"))

# mirror of `pstat_params` (include/pstat.h) -- field order and types must match
struct PstatParams
  E0::Cdouble; K1::Cdouble; K2::Cdouble; mu::Cdouble; kT::Cdouble; Fz::Cdouble; Fx::Cdouble; b::Cdouble
  phi_step::Cdouble; theta_step::Cdouble
  adj_lb::Cdouble; adj_ub::Cdouble; adj_scale::Cdouble
  steps_per_adjust::Int64; n::Int64; num_chains::Int64
  seed::UInt64; chain_id0::UInt64
  chain_type::Int32; energy_type::Int32; do_flips::Int32; umbrella::Int32; precision::Int32; device::Int32
  rng::Int32; move_set::Int32
  bend_mod::Cdouble; bend_angle::Cdouble; cluster_prob::Cdouble
  x0_phi::Cdouble; x0_theta::Cdouble; dx0_phi::Cdouble; dx0_theta::Cdouble
  use_x0::Int32; uniform_bits::Int32      # 0 = the precision's default (53 random bits in the Metropolis eps for f64)
  cutoff_radius::Cdouble
end

# mirror of `pstat_summary`
struct PstatSummary
  avg::NTuple{16,Cdouble}; stderr::NTuple{16,Cdouble}
  acceptance_ratio::Cdouble; ar_stderr::Cdouble
  num_chains::Int64; steps_per_chain::Int64; attempted_updates::Cdouble
  extra_avg::NTuple{2,Cdouble}; extra_stderr::NTuple{2,Cdouble}
  nan_rejects::Int64; chains_collapsed::Int64
end

const NQ = 19                   # PSTAT_NQ
const NRED = 1 + 2 * NQ + 2     # PSTAT_NRED

function check(rc::Cint)
  if rc != 0
    what = unsafe_string(ccall((:pstat_strerror, LIBPSTAT), Cstring, (Cint,), rc))
    detail = unsafe_string(ccall((:pstat_last_error, LIBPSTAT), Cstring, ()))
    error("libpstat: $what ($rc): $detail")
  end
end

s = ArgParseSettings();
@add_arg_table! s begin   # the reference's table, mcmc_clustering_eap_chain.jl:14-152
  "--E0", "-e";            arg_type = Float64; default = 0.0;  help = "magnitude of electric field"
  "--chain-type", "-T";    arg_type = String;  default = "dielectric"; help = "chain type (dielectric|polar)"
  "--K1", "-J";            arg_type = Float64; default = 1.0
  "--K2", "-K";            arg_type = Float64; default = 0.0
  "--mu", "-m";            arg_type = Float64; default = 1e-2
  "--bend-mod", "-a";      arg_type = Float64; default = 0.0;  help = "bending modulus of chain"
  "--bend-angle", "-g";    arg_type = Float64; default = 0.0;  help = "zero energy bond angle"
  "--energy-type", "-u";   arg_type = String;  default = "Ising"; help = "energy type (interacting|cutoff|Ising|noninteracting)"
  "--cutoff-radius";       arg_type = Float64; default = 7.5
  "--kT", "-k";            arg_type = Float64; default = 1.0
  "--Fz", "-F";            arg_type = Float64; default = 0.0
  "--Fx", "-G";            arg_type = Float64; default = 0.0
  "--mlen", "-b";          arg_type = Float64; default = 1.0
  "--num-monomers", "-n";  arg_type = Int;     default = 100
  "--num-steps", "-N";     arg_type = Int;     default = convert(Int, 1e6)
  "--phi-step", "-p";      arg_type = Float64; default = 3*π / 8
  "--theta-step", "-q";    arg_type = Float64; default = 3*π / 16
  "--cluster-prob";        arg_type = Float64; default = 0.5
  "--step-adjust-lb", "-L"; arg_type = Float64; default = 0.15
  "--step-adjust-ub", "-U"; arg_type = Float64; default = 0.40
  "--step-adjust-scale", "-A"; arg_type = Float64; default = 1.1
  "--steps-per-adjust", "-S"; arg_type = Int; default = 2500
  "--umbrella-sampling", "-B"; action = :store_true
  "--update-freq";         arg_type = Float64; default = 15.0
  "--verbose", "-v";       arg_type = Int;     default = 3
  "--prefix", "-P";        arg_type = String;  default = "eap-mcmc"
  "--postfix", "-Q";       arg_type = String;  default = ""
  "--stepout", "-s";       arg_type = Int;     default = 500
  "--numeric-type";        arg_type = String;  default = "float64"
  "--burn-in";             arg_type = Int;     default = 50000
  "--burn-schedule";       arg_type = String;  default = "[1000; 100; 10; 2; 1]"
  "--x0";                  arg_type = String
  "--dx0";                 arg_type = String;  default = "[2*pi, 1e-1]"
  "--profile", "-Z";       action = :store_true
  # added by this implementation
  "--num-chains";          arg_type = Int;     default = 4096
  "--seed";                arg_type = Int;     default = -1;  help = "seed of the per-chain generators; default (-1): fresh OS entropy per run, like the reference's unseeded RNG"
  "--devices";             arg_type = String;  default = "0"
  "--precision";           arg_type = String;  default = "f64";  help = "device arithmetic: f64 (the reference's Float64) | f32 (fast path) | q16"
  "--rng";                 arg_type = String;  default = "mwc64x"
  "--uniform-bits";        arg_type = Int;     default = 0;   help = "random bits of the Metropolis draw rand(): 0 = the precision's default (53 for f64, like Julia's Float64 rand(); 23 for f32 / q16) | 23 | 53"
end

pargs = parse_args(s);
# The reference never seeds Julia's RNG: the same command line launched 25 times gives 25 independent samples
# (run/interacting-compare-with-clustering_2021-09-28.jl:26-27).  Same here unless --seed is given.
const SEED_WAS_DRAWN = pargs["seed"] < 0
if SEED_WAS_DRAWN
  pargs["seed"] = Int(rand(RandomDevice(), UInt64) >> 1)
end

if pargs["verbose"] == 3
  global_logger(ConsoleLogger(stderr, Logging.Info));
elseif pargs["verbose"] == 2
  global_logger(ConsoleLogger(stderr, Logging.Warn));
elseif pargs["verbose"] == 1
  global_logger(ConsoleLogger(stderr, Logging.Error));
else
  global_logger(Logging.NullLogger());
end
SEED_WAS_DRAWN && pargs["verbose"] >= 2 &&
  println(stderr, "[ Info: seed: $(pargs["seed"]) (fresh entropy; pass --seed $(pargs["seed"]) to reproduce this run)");

# --x0 / --dx0 exactly as EAPChain(pargs) reads them (inc/eap_chain.jl:61-79)
function start_configuration(pargs)
  (!haskey(pargs, "x0") || isnothing(pargs["x0"])) && return (nothing, nothing)
  x0 = eval(Meta.parse(pargs["x0"]))
  dx0 = eval(Meta.parse(pargs["dx0"]))
  if !(typeof(x0) <: Vector || typeof(dx0) <: Vector)
    error("Invalid input for 'x0' and/or 'dx0', $(pargs["x0"]); $(pargs["dx0"])")
  end
  (length(x0) == 2 || length(x0) == 2*pargs["num-monomers"]) || error("Invalid input for 'x0', $(pargs["x0"])")
  return (Vector{Cdouble}(x0), Vector{Cdouble}(dx0))
end

function params(pargs, num_chains, chain_id0, device, x0, dx0)
  ct = get(Dict("dielectric" => 0, "polar" => 1), pargs["chain-type"], -1)
  ct >= 0 || error("chain-type is not understood.")
  et = get(Dict("noninteracting" => 0, "interacting" => 1, "Ising" => 2, "cutoff" => 3), pargs["energy-type"], -1)
  et >= 0 || error("energy-type is not understood.")
  prec = get(Dict("f32" => 0, "f64" => 1, "q16" => 2), pargs["precision"], -1)
  prec >= 0 || error("precision '$(pargs["precision"])' not understood")
  rng = get(Dict("mwc64x" => 0, "xoshiro128++" => 1), pargs["rng"], -1)
  rng >= 0 || error("rng '$(pargs["rng"])' not understood")
  uniform_x0 = !isnothing(x0) && length(x0) == 2
  PstatParams(pargs["E0"], pargs["K1"], pargs["K2"], pargs["mu"], pargs["kT"], pargs["Fz"], pargs["Fx"],
              pargs["mlen"], pargs["phi-step"], pargs["theta-step"], pargs["step-adjust-lb"],
              pargs["step-adjust-ub"], pargs["step-adjust-scale"], pargs["steps-per-adjust"],
              pargs["num-monomers"], num_chains, UInt64(pargs["seed"]), UInt64(chain_id0),
              ct, et, 0, pargs["umbrella-sampling"] ? 1 : 0, prec, device, rng,
              1,                                    # move_set = PSTAT_MOVES_CLUSTER
              pargs["bend-mod"], pargs["bend-angle"], pargs["cluster-prob"],
              uniform_x0 ? x0[1] : 0.0, uniform_x0 ? x0[2] : 0.0,
              isnothing(dx0) ? 2pi : dx0[1], isnothing(dx0) ? 0.1 : dx0[2],
              uniform_x0 ? 1 : 0, pargs["uniform-bits"], pargs["cutoff-radius"])
end

# --numeric-type (mcmc_eap_chain.jl:186-197): the per-chain sums are Float64 on the device (the reference's default);
# the option selects the type in which the per-chain means are merged.  Float128 / Dec128 need Quadmath / DecFP,
# as in the reference.
function wide_type(name)
  name == "float64" && return Float64
  name == "big" && return BigFloat
  if name == "float128"
    @eval using Quadmath
    return Base.invokelatest(() -> Quadmath.Float128)
  end
  @eval using DecFP
  return Base.invokelatest(() -> DecFP.Dec128)
end

# pooled mean and across-chain standard error of the NQ per-chain running means, in type T
function wide_merge(handles, num_chains_of, T)
  cols = Vector{Matrix{Float64}}()
  for (h, m) in zip(handles, num_chains_of)
    buf = zeros(Cdouble, m, NQ)      # column-major: [chain, quantity] = out[q * nchains + k]
    check(ccall((:pstat_chain_means, LIBPSTAT), Cint, (Ptr{Cvoid}, Int32, Ptr{Cdouble}), h, -1, buf))
    push!(cols, buf)
  end
  m = vcat(cols...)
  C = size(m, 1)
  avg = [sum(T.(m[:, q])) / C for q = 1:NQ]
  se = [C > 1 ? sqrt(sum((T.(m[:, q]) .- avg[q]) .^ 2) / (C - 1) / C) : zero(T) for q = 1:NQ]
  return avg, se
end

function report_failures(sm)
  sm.nan_rejects > 0 &&
    @warn "$(sm.nan_rejects) proposals had a non-finite energy and were rejected";
  sm.chains_collapsed > 0 &&
    @warn "$(sm.chains_collapsed) of $(sm.num_chains) chains have collapsed (|U| a thousand times beyond field + force + thermal energy: monomers on top of each other)";
end

function pooled_summary(handles, steps)
  red = zeros(Cdouble, NRED)
  tmp = zeros(Cdouble, NRED)
  for h in handles
    check(ccall((:pstat_reduce_host, LIBPSTAT), Cint, (Ptr{Cvoid}, Int32, Ptr{Cdouble}), h, -1, tmp))
    red .+= tmp
  end
  out = Ref{PstatSummary}()
  check(ccall((:pstat_summary_from_reduction, LIBPSTAT), Cint, (Ptr{Cdouble}, Int64, Ref{PstatSummary}),
              red, steps, out))
  return out[]
end

# replaces EAPChain(pargs) (mcmc_clustering_eap_chain.jl:167-170): one handle per device, chains sharded by id
function create_chains(pargs)
  (x0, dx0) = start_configuration(pargs)
  devices = [parse(Int, d) for d in split(pargs["devices"], ",") if d != ""]
  total = pargs["num-chains"]
  handles = Ptr{Cvoid}[]
  first = 0
  for (i, dev) in enumerate(devices)
    cnt = div(total, length(devices)) + (i <= rem(total, length(devices)) ? 1 : 0)
    cnt == 0 && continue
    p = Ref(params(pargs, cnt, first, dev, x0, dx0))
    h = Ref{Ptr{Cvoid}}(C_NULL)
    check(ccall((:pstat_create, LIBPSTAT), Cint, (Ref{PstatParams}, Int32, Ptr{Cvoid}, Ref{Ptr{Cvoid}}),
                p, 1, C_NULL, h))
    if !isnothing(x0) && length(x0) != 2      # per-monomer start, inc/eap_chain.jl:73-75
      check(ccall((:pstat_restart_from_x0, LIBPSTAT), Cint, (Ptr{Cvoid}, Ptr{Cdouble}, Int64, Cdouble, Cdouble),
                  h[], x0, length(x0), dx0[1], dx0[2]))
    end
    push!(handles, h[])
    first += cnt
  end
  return handles
end

# dipoles of the printed microstate, for the trajectory file only (inc/dipole_response.jl:7-29)
function dipoles(pargs, ϕs, θs)
  n̂s = hcat([[cos(ϕ)*sin(θ), sin(ϕ)*sin(θ), cos(θ)] for (ϕ, θ) in zip(ϕs, θs)]...)
  if pargs["chain-type"] == "dielectric"
    a = (pargs["K1"] - pargs["K2"]) * pargs["E0"]
    return hcat([a*n̂s[3, i]*n̂s[:, i] + [0.0, 0.0, pargs["K2"]*pargs["E0"]] for i in 1:size(n̂s, 2)]...)
  end
  return pargs["mu"] * n̂s
end

# one call of the reference's mcmc(nsteps, pargs, chain) (:172-352): fresh acceptor, step sizes, averagers
function mcmc(nsteps::Int, pargs, handles, kT; write_files::Bool)
  pargs["numeric-type"] in ("float64", "float128", "dec128", "big") ||
    error("numeric-type '$(pargs["numeric-type"])' not understood");
  n = pargs["num-monomers"]
  for h in handles
    check(ccall((:pstat_set_kT, LIBPSTAT), Cint, (Ptr{Cvoid}, Int32, Cdouble), h, -1, kT))
    check(ccall((:pstat_reset_sampler, LIBPSTAT), Cint, (Ptr{Cvoid},), h))
    check(ccall((:pstat_reset_averages, LIBPSTAT), Cint, (Ptr{Cvoid},), h))
  end
  outfile = rollfile = nothing
  if write_files   # every mcmc() call of the reference rewrites the files: only the last call's survive
    outfile = open("$(pargs["prefix"])_trajectory.csv", "w");
    println(outfile, join(vcat(["step", "r1", "r2", "r3", "p1", "p2", "p3", "U"],
                               vcat([["phi$i", "theta$i"] for i=1:n]...),
                               vcat([["mux$i", "muy$i", "muz$i"] for i=1:n]...)), ","));
    rollfile = open("$(pargs["prefix"])_rolling.csv", "w");
    println(rollfile, "step,r1,r2,r3,r1sq,r2sq,r3sq,rsq,p1,p2,p3,p1sq,p2sq,p3sq,psq,U,Usq,Ealign,psi");
  end
  start = time(); last_update = start
  stepout = pargs["stepout"]
  step = 0
  while step < nsteps
    seg = nsteps - step
    if write_files && stepout > 0; seg = min(seg, stepout - step % stepout); end
    for h in handles   # asynchronous: the devices run concurrently
      check(ccall((:pstat_advance, LIBPSTAT), Cint, (Ptr{Cvoid}, Int64), h, seg))
    end
    step += seg
    if time() - last_update > pargs["update-freq"]
      @info "elapsed: $(time() - start)";
      @info "step:    $step / $nsteps";
      last_update = time();
    end
    if write_files && stepout > 0 && step % stepout == 0
      micro = zeros(Cdouble, 7)
      check(ccall((:pstat_microstate, LIBPSTAT), Cint, (Ptr{Cvoid}, Int64, Ptr{Cdouble}), handles[1], 0, micro))
      angles = zeros(Cdouble, 2n)       # theta_1..n, then phi_1..n
      check(ccall((:pstat_chain_state, LIBPSTAT), Cint,
                  (Ptr{Cvoid}, Int64, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Int64}, Ptr{Cdouble}, Ptr{UInt32}),
                  handles[1], 0, angles, C_NULL, C_NULL, C_NULL, C_NULL))
      θs = angles[1:n]; ϕs = angles[n+1:2n]
      sm = pooled_summary(handles, step)
      println(outfile, join(string.(vcat(Float64(step), micro, reshape(vcat(transpose(ϕs), transpose(θs)), :),
                                         reshape(dipoles(pargs, ϕs, θs), :))), ","))
      println(rollfile, join(string.(vcat(Float64(step), collect(sm.avg), collect(sm.extra_avg))), ","))
    end
  end
  sm = pooled_summary(handles, nsteps)
  @info "total time elapsed: $(time() - start)";
  @info "acceptance rate: $(sm.acceptance_ratio)";
  report_failures(sm);
  if write_files; close(outfile); close(rollfile); end
  return sm
end

sm = if pargs["profile"]
  error("Not currently implemented...");
else
  handles = create_chains(pargs)
  kT_multipliers = eval(Meta.parse(pargs["burn-schedule"]));    # :365
  for kT_mult in kT_multipliers                                  # :366-383
    mcmc(pargs["burn-in"], pargs, handles, pargs["kT"] * kT_mult; write_files = false)
  end
  result = mcmc(pargs["num-steps"], pargs, handles, pargs["kT"]; write_files = true)   # :385-386
  if pargs["numeric-type"] != "float64"
    T = wide_type(pargs["numeric-type"])
    @warn "--numeric-type $(pargs["numeric-type"]): per-chain sums are Float64 on the device; the merge over chains is carried out in $T";
    counts = Int[]
    for h in handles   # chains held by each handle: entry [0] of its reduction vector
      tmp = zeros(Cdouble, NRED)
      check(ccall((:pstat_reduce_host, LIBPSTAT), Cint, (Ptr{Cvoid}, Int32, Ptr{Cdouble}), h, -1, tmp))
      push!(counts, Int(round(tmp[1])))
    end
    (wmean, _) = Base.invokelatest(wide_merge, handles, counts, T)   # (T's methods may come from a package loaded just now)
    global WIDE = wmean
  end
  for h in handles
    ccall((:pstat_destroy, LIBPSTAT), Cvoid, (Ptr{Cvoid},), h)
  end
  result
end

a = collect(sm.avg); x = collect(sm.extra_avg); ar = sm.acceptance_ratio
if @isdefined WIDE
  a = WIDE[1:16]; ar = WIDE[17]; x = WIDE[18:19]
end
println("<r>    =   $(a[1:3])");
println("<r/nb> =   $(a[1:3] / (pargs["mlen"]*pargs["num-monomers"]))");
println("<rj2>  =   $(a[4:6])");
println("<r2>   =   $(a[7])");
println("<p>    =   $(a[8:10])");
println("<pj2>  =   $(a[11:13])");
println("<p2>   =   $(a[14])");
println("<U>    =   $(a[15])");
println("<U2>   =   $(a[16])");
println("<cos2(θ)>   =   $(x[1])");
println("<ψ>    =   $(x[2])");
println("AR     =   $ar");
