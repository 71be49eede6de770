# mcmc_eap_chain.jl -- Julia host of the MI355X fixed-force MCMC path.
#
# Same command line, same files, same ten stdout lines as the reference's mcmc_eap_chain.jl; the
# step loop runs on the GPU through libpstat (C ABI, include/pstat.h) via ccall.
# NOT EXECUTED IN THE BUILD IMAGE (no Julia toolchain there); it is the twin of
# polymer_stats_amd/mcmc_eap_chain.py, which the test-suite exercises.  Only ArgParse is needed.
#
#   julia julia/mcmc_eap_chain.jl --chain-type dielectric -n 100 -e 1 -F 1 -N 100000 \
#         --num-chains 65536 --prefix out/run1 -v 2
using ArgParse
using Logging
using Random
using Printf

const LIBPSTAT = get(ENV, "PSTAT_LIB", joinpath(@__DIR__, "..", "polymer_stats_amd", "libpstat.so"))
const NOBS = 16

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
@add_arg_table! s begin
  "--E0", "-e";            arg_type = Float64; default = 0.0;  help = "magnitude of electric field"
  "--chain-type", "-T";    arg_type = String;  default = "dielectric"; help = "chain type (dielectric|polar)"
  "--K1", "-J";            arg_type = Float64; default = 1.0
  "--K2", "-K";            arg_type = Float64; default = 0.0
  "--mu", "-m";            arg_type = Float64; default = 1e-2
  "--energy-type", "-u";   arg_type = String;  default = "noninteracting"
  "--kT", "-k";            arg_type = Float64; default = 1.0
  "--ensemble-type", "-E"; arg_type = String;  default = "force"
  "--Fz", "-F";            arg_type = Float64; default = 0.0
  "--Fx", "-G";            arg_type = Float64; default = 0.0
  "--rz", "-z";            arg_type = Float64; default = 0.0
  "--rx", "-x";            arg_type = Float64; default = 0.0
  "--mlen", "-b";          arg_type = Float64; default = 1.0
  "--num-monomers", "-n";  arg_type = Int;     default = 100
  "--num-steps", "-N";     arg_type = Int;     default = convert(Int, 1e5)
  "--num-inits", "-M";     arg_type = Int;     default = 1
  "--force-init", "-I";    action = :store_true
  "--phi-step", "-p";      arg_type = Float64; default = 3*π / 8
  "--do-flips";            action = :store_true
  "--theta-step", "-q";    arg_type = Float64; default = 3*π / 16
  "--chain-frac-step", "-f"; arg_type = Float64; default = 0.15
  "--step-adjust-lb", "-L"; arg_type = Float64; default = 0.15
  "--step-adjust-ub", "-U"; arg_type = Float64; default = 0.55
  "--step-adjust-scale", "-A"; arg_type = Float64; default = 1.1
  "--steps-per-adjust", "-S"; arg_type = Int; default = 2500
  "--acc", "-a";           arg_type = String;  default = "metropolis"
  "--umbrella-sampling", "-B"; action = :store_true
  "--update-freq";         arg_type = Float64; default = 15.0
  "--verbose", "-v";       arg_type = Int;     default = 3
  "--prefix", "-P";        arg_type = String;  default = "eap-mcmc"
  "--postfix", "-Q";       arg_type = String;  default = ""
  "--stepout", "-s";       arg_type = Int;     default = 500
  "--numeric-type";        arg_type = String;  default = "float64"
  "--profile", "-Z";       action = :store_true
  # added by this implementation
  "--num-chains";          arg_type = Int;     default = 4096
  "--seed";                arg_type = Int;     default = -1;  help = "seed of the per-chain generators; default (-1): fresh OS entropy per run, like the reference's unseeded RNG"
  "--devices";             arg_type = String;  default = "0"
  "--precision";           arg_type = String;  default = "f64";  help = "device arithmetic: f64 (the reference's Float64) | f32 (fast path) | q16"
  "--rng";                 arg_type = String;  default = "mwc64x"
  "--uniform-bits";        arg_type = Int;     default = 0;   help = "random bits of the Metropolis draw rand(): 0 = the precision's default (53 for f64, like Julia's Float64 rand(); 23 for f32 / q16) | 23 | 53"
  "--burn-in";             arg_type = Int;     default = 0
  "--burn-schedule";       arg_type = String;  default = "[1]"
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

function params(pargs, num_chains, chain_id0, device)
  ct = get(Dict("dielectric" => 0, "polar" => 1), pargs["chain-type"], -1)
  ct >= 0 || error("chain-type is not understood.")
  et = get(Dict("noninteracting" => 0, "interacting" => 1, "Ising" => 2), pargs["energy-type"], -1)
  et >= 0 || error("energy-type is not understood.")
  prec = get(Dict("f32" => 0, "f64" => 1, "q16" => 2), pargs["precision"], -1)
  prec >= 0 || error("precision '$(pargs["precision"])' not understood")
  rng = get(Dict("mwc64x" => 0, "xoshiro128++" => 1), pargs["rng"], -1)
  rng >= 0 || error("rng '$(pargs["rng"])' not understood")
  PstatParams(pargs["E0"], pargs["K1"], pargs["K2"], pargs["mu"], pargs["kT"], pargs["Fz"], pargs["Fx"],
              pargs["mlen"], pargs["phi-step"], pargs["theta-step"], pargs["step-adjust-lb"],
              pargs["step-adjust-ub"], pargs["step-adjust-scale"], pargs["steps-per-adjust"],
              pargs["num-monomers"], num_chains, UInt64(pargs["seed"]), UInt64(chain_id0),
              ct, et, pargs["do-flips"] ? 1 : 0, pargs["umbrella-sampling"] ? 1 : 0, prec, device, rng,
              0,                                   # move_set = PSTAT_MOVES_SINGLE: this main
              0.0, 0.0, 0.5, 0.0, 0.0, 2pi, 0.1,   # clustering-main options at their defaults (unused here)
              0, pargs["uniform-bits"], 7.5)
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

function mcmc(nsteps::Int, pargs)
  pargs["acc"] == "metropolis" ||
    error("'$(pargs["acc"])' acceptance criteria has not yet been implemented.");
  pargs["numeric-type"] in ("float64", "float128", "dec128", "big") ||
    error("numeric-type '$(pargs["numeric-type"])' not understood");
  pargs["ensemble-type"] == "force" ||
    error("'end-to-end' ensemble is an experimental option of the reference; it has no device implementation");

  devices = [parse(Int, d) for d in split(pargs["devices"], ",") if d != ""]
  total = pargs["num-chains"]
  handles = Ptr{Cvoid}[]
  counts = Int[]
  first = 0
  for (i, dev) in enumerate(devices)
    cnt = div(total, length(devices)) + (i <= rem(total, length(devices)) ? 1 : 0)
    cnt == 0 && continue
    p = Ref(params(pargs, cnt, first, dev))
    h = Ref{Ptr{Cvoid}}(C_NULL)
    check(ccall((:pstat_create, LIBPSTAT), Cint, (Ref{PstatParams}, Int32, Ptr{Cvoid}, Ref{Ptr{Cvoid}}),
                p, 1, C_NULL, h))
    push!(handles, h[])
    push!(counts, cnt)
    first += cnt
  end

  outfile = open("$(pargs["prefix"])_trajectory.csv", "w");
  println(outfile, "step,r1,r2,r3,p1,p2,p3,U");
  rollfile = open("$(pargs["prefix"])_rolling.csv", "w");
  println(rollfile, "step,r1,r2,r3,r1sq,r2sq,r3sq,rsq,p1,p2,p3,p1sq,p2sq,p3sq,psq,U,Usq");

  if pargs["burn-in"] > 0   # temperature ladder whose records are discarded
    for mult in eval(Meta.parse(pargs["burn-schedule"]))
      for h in handles
        check(ccall((:pstat_set_kT, LIBPSTAT), Cint, (Ptr{Cvoid}, Int32, Cdouble), h, -1, pargs["kT"] * mult))
        check(ccall((:pstat_advance, LIBPSTAT), Cint, (Ptr{Cvoid}, Int64), h, pargs["burn-in"]))
      end
    end
    for h in handles
      check(ccall((:pstat_set_kT, LIBPSTAT), Cint, (Ptr{Cvoid}, Int32, Cdouble), h, -1, pargs["kT"]))
      check(ccall((:pstat_reset_averages, LIBPSTAT), Cint, (Ptr{Cvoid},), h))
    end
  end
  start = time(); last_update = start; recorded = 0
  stepout = pargs["stepout"]
  for init = 1:pargs["num-inits"]
    step = 0
    while step < nsteps
      seg = nsteps - step
      if stepout > 0; seg = min(seg, stepout - step % stepout); end
      for h in handles   # asynchronous: the devices run concurrently
        check(ccall((:pstat_advance, LIBPSTAT), Cint, (Ptr{Cvoid}, Int64), h, seg))
      end
      step += seg; recorded += seg
      if time() - last_update > pargs["update-freq"]
        @info "elapsed: $(time() - start)";
        @info "init:    $init / $(pargs["num-inits"])";
        @info "step:    $step / $nsteps";
        last_update = time();
      end
      if stepout > 0 && step % stepout == 0
        micro = zeros(Cdouble, 7)
        check(ccall((:pstat_microstate, LIBPSTAT), Cint, (Ptr{Cvoid}, Int64, Ptr{Cdouble}), handles[1], 0, micro))
        sm = pooled_summary(handles, recorded)
        println(outfile, join(string.(vcat(Float64(step), micro)), ","))
        println(rollfile, join(string.(vcat(Float64(step), collect(sm.avg))), ","))
      end
    end
    if init < pargs["num-inits"]
      for h in handles
        check(ccall((:pstat_reinit, LIBPSTAT), Cint, (Ptr{Cvoid}, Int32), h, pargs["force-init"] ? 1 : 0))
      end
    end
  end
  sm = pooled_summary(handles, recorded)
  @info "total time elapsed: $(time() - start)";
  @info "acceptance rate: $(sm.acceptance_ratio)";
  report_failures(sm);
  close(outfile); close(rollfile);
  a = collect(sm.avg)
  ar = sm.acceptance_ratio
  if pargs["numeric-type"] != "float64"
    T = wide_type(pargs["numeric-type"])
    @warn "--numeric-type $(pargs["numeric-type"]): per-chain sums are Float64 on the device; the merge over chains is carried out in $T";
    (wmean, _) = Base.invokelatest(wide_merge, handles, counts, T)   # (T's methods may come from a package loaded just now)
    a = wmean[1:NOBS]; ar = wmean[NOBS + 1]
  end
  for h in handles
    ccall((:pstat_destroy, LIBPSTAT), Cvoid, (Ptr{Cvoid},), h)
  end
  # (scalar averages r2, p2, U, U2), (vector averages r, rj2, p, pj2), acceptance ratio
  return ([a[7], a[14], a[15], a[16]], [a[1:3], a[4:6], a[8:10], a[11:13]], ar)
end

(sas, vas, ar) = if pargs["profile"]
  error("not implemented for the HPC env");
else
  mcmc(pargs["num-steps"], pargs);
end

println("<r>    =   $(vas[1])");
println("<r/nb> =   $(vas[1] / (pargs["mlen"]*pargs["num-monomers"]))");
println("<rj2>  =   $(vas[2])");
println("<r2>   =   $(sas[1])");
println("<p>    =   $(vas[3])");
println("<pj2>  =   $(vas[4])");
println("<p2>   =   $(sas[2])");
println("<U>    =   $(sas[3])");
println("<U2>   =   $(sas[4])");
println("AR     =   $ar");
