# PenguinHIP.jl -- drop-in replacement of Penguin.jl's hot path
#     Mesh -> Capacity -> DiffusionOps -> Phase -> DiffusionUnsteadyMono -> solve_DiffusionUnsteadyMono!
# on AMD MI355X: same exported names, positional/keyword signatures and field names as Penguin.jl
# (src/Penguin.jl:25-75), every numeric operation done by libpenguin_hip.so through `ccall`
# (C ABI: include/penguin_hip.h).
#
# NOTE: there is no Julia in the authoring container, so this file has never been executed; it is the
# reference-side binding a maintainer would add (INTEGRATION.md).  The Python mirror
# penguin/jl_amd/api.py binds the SAME symbols with ctypes and is what the test-suite drives.
module PenguinHIP

using SparseArrays, StaticArrays

export Mesh, nC, Capacity, Sphere, MultiSphere, DiffusionOps, Phase, Dirichlet, Neumann, Robin, Periodic,
       ScalarJump, FluxJump, BorderConditions, InterfaceConditions, Solver, DiffusionUnsteadyMono,
       solve_DiffusionUnsteadyMono!, ∇, ∇₋

const libpg = get(ENV, "PENGUIN_HIP_LIB", joinpath(@__DIR__, "..", "penguin", "jl_amd", "lib", "libpenguin_hip.so"))

function check(status::Int32)
    if status != 0
        buf = Vector{UInt8}(undef, 4096)
        ccall((:pg_last_error, libpg), Int32, (Ptr{UInt8}, Csize_t), buf, length(buf))
        error(unsafe_string(pointer(buf)))          # Penguin.jl raises with error(...) too
    end
    nothing
end

const _initialised = Ref(false)
function init(device::Integer=0)
    _initialised[] && return
    check(ccall((:pg_init, libpg), Int32, (Int32,), device))
    _initialised[] = true
end

# ---------------------------------------------------------------------------------- Mesh  (src/mesh.jl:41-79)
struct MeshTag{N}
    border_cells::Vector{Tuple{CartesianIndex{N}, NTuple{N, Float64}}}
end
abstract type AbstractMesh end
mutable struct Mesh{N} <: AbstractMesh
    centers::NTuple{N, Vector{Float64}}
    nodes::NTuple{N, Vector{Float64}}
    tag::MeshTag
    dims::NTuple{N, Int}
    handle::Ptr{Cvoid}
end
function Mesh(n::NTuple{N, Int}, domain_size::NTuple{N, Float64}, x0::NTuple{N, Float64}=ntuple(_ -> 0.0, N)) where N
    h = Ref{Ptr{Cvoid}}(C_NULL)
    nv, Lv, xv = collect(Int64, n), collect(domain_size), collect(x0)
    check(ccall((:pg_mesh_create, libpg), Int32, (Int32, Ptr{Int64}, Ptr{Float64}, Ptr{Float64}, Ptr{Ptr{Cvoid}}),
                N, nv, Lv, xv, h))
    centers = ntuple(d -> begin v = Vector{Float64}(undef, n[d])
        check(ccall((:pg_mesh_get_centers, libpg), Int32, (Ptr{Cvoid}, Int32, Ptr{Float64}, Int64), h[], d - 1, v, n[d])); v end, N)
    nodes = ntuple(d -> begin v = Vector{Float64}(undef, n[d] + 1)
        check(ccall((:pg_mesh_get_nodes, libpg), Int32, (Ptr{Cvoid}, Int32, Ptr{Float64}, Int64), h[], d - 1, v, n[d] + 1)); v end, N)
    nb = Ref{Int64}(0)
    check(ccall((:pg_mesh_num_border_cells, libpg), Int32, (Ptr{Cvoid}, Ptr{Int64}), h[], nb))
    idx = Matrix{Int64}(undef, N, nb[]); pos = Matrix{Float64}(undef, N, nb[]); key = Vector{Int32}(undef, nb[])
    check(ccall((:pg_mesh_get_border_cells, libpg), Int32, (Ptr{Cvoid}, Ptr{Int64}, Ptr{Float64}, Ptr{Int32}), h[], idx, pos, key))
    border = [(CartesianIndex(ntuple(d -> Int(idx[d, q]), N)), ntuple(d -> pos[d, q], N)) for q in 1:nb[]]
    m = Mesh{N}(centers, nodes, MeshTag{N}(border), n, h[])
    finalizer(x -> ccall((:pg_mesh_destroy, libpg), Int32, (Ptr{Cvoid},), x.handle), m)
    m
end
nC(mesh::AbstractMesh) = prod(mesh.dims)

# ---------------------------------------------------------------------------------- bodies
# A tagged level set the GPU can evaluate; calling it gives the same signed distance a Penguin.jl closure would.
struct Sphere{N} <: Function
    center::NTuple{N, Float64}
    radius::Float64
    complement::Bool
end
Sphere(center::NTuple{N, Float64}, radius::Float64; complement::Bool=false) where N = Sphere{N}(center, radius, complement)
(s::Sphere{N})(x...) where N = (f = sqrt(sum((x[d] - s.center[d])^2 for d in 1:N)) - s.radius; s.complement ? -f : f)

# ---------------------------------------------------------------------------------- Capacity (src/capacity.jl:25-36)
abstract type AbstractCapacity end
mutable struct Capacity{N} <: AbstractCapacity
    A::NTuple{N, SparseMatrixCSC{Float64, Int}}
    B::NTuple{N, SparseMatrixCSC{Float64, Int}}
    V::SparseMatrixCSC{Float64, Int}
    W::NTuple{N, SparseMatrixCSC{Float64, Int}}
    C_ω::Vector{SVector{N, Float64}}
    C_γ::Vector{SVector{N, Float64}}
    Γ::SparseMatrixCSC{Float64, Int}
    cell_types::Vector{Float64}
    mesh::AbstractMesh
    body::Function
    handle::Ptr{Cvoid}
end
const PG_CAP_V, PG_CAP_GAMMA, PG_CAP_CELL_TYPES, PG_CAP_A, PG_CAP_B, PG_CAP_W, PG_CAP_C_OMEGA, PG_CAP_C_GAMMA = 0:7
function _field(h, field, d, M)
    v = zeros(M)
    check(ccall((:pg_capacity_get, libpg), Int32, (Ptr{Cvoid}, Int32, Int32, Ptr{Float64}, Int64), h, field, d, v, M)); v
end
function Capacity(body::Sphere{N}, mesh::Mesh{N}; method::String="VOFI", compute_centroids::Bool=true) where N
    init()
    params = vcat(collect(body.center), body.radius)
    flags = Int32((body.complement ? 1 : 0) | (compute_centroids ? 0 : 2))
    h = Ref{Ptr{Cvoid}}(C_NULL)
    check(ccall((:pg_capacity_create_levelset, libpg), Int32, (Ptr{Cvoid}, Int32, Ptr{Float64}, Int32, Int32, Ptr{Ptr{Cvoid}}),
                mesh.handle, 1, params, length(params), flags, h))
    M = prod(mesh.dims .+ 1)
    diagm(v) = spdiagm(0 => v)
    A = ntuple(d -> diagm(_field(h[], PG_CAP_A, d - 1, M)), N)
    B = ntuple(d -> diagm(_field(h[], PG_CAP_B, d - 1, M)), N)
    W = ntuple(d -> diagm(_field(h[], PG_CAP_W, d - 1, M)), N)
    cw = [_field(h[], PG_CAP_C_OMEGA, d - 1, M) for d in 1:N]
    C_ω = [SVector{N, Float64}(ntuple(d -> cw[d][i], N)) for i in 1:M]
    C_γ = if compute_centroids
        cg = [_field(h[], PG_CAP_C_GAMMA, d - 1, M) for d in 1:N]
        [SVector{N, Float64}(ntuple(d -> cg[d][i], N)) for i in 1:M]
    else
        Vector{SVector{N, Float64}}(undef, 0)
    end
    c = Capacity{N}(A, B, diagm(_field(h[], PG_CAP_V, 0, M)), W, C_ω, C_γ, diagm(_field(h[], PG_CAP_GAMMA, 0, M)),
                    _field(h[], PG_CAP_CELL_TYPES, 0, M), mesh, body, h[])
    finalizer(x -> ccall((:pg_capacity_destroy, libpg), Int32, (Ptr{Cvoid},), x.handle), c)
    c
end
# arbitrary Julia closures: compute the capacities with Penguin.jl's own VOFI path and hand the arrays over
function Capacity(body::Function, mesh::Mesh{N}; kwargs...) where N
    error("PenguinHIP.Capacity: arbitrary level-set closures cannot run on the GPU; pass a Sphere, or build the " *
          "capacity with Penguin.Capacity and call PenguinHIP.capacity_from_arrays(penguin_capacity, mesh)")
end

# ---------------------------------------------------------------------------------- DiffusionOps (src/operators.jl:49-55)
abstract type AbstractOperators end
mutable struct DiffusionOps{N} <: AbstractOperators
    G::SparseMatrixCSC{Float64, Int}
    H::SparseMatrixCSC{Float64, Int}
    Wꜝ::SparseMatrixCSC{Float64, Int}
    V::SparseMatrixCSC{Float64, Int}
    size::NTuple{N, Int}
    handle::Ptr{Cvoid}
end
function _export_csc(h, which, nrows, ncols)
    nnz = Ref{Int64}(0)
    check(ccall((:pg_diffops_export_csc, libpg), Int32, (Ptr{Cvoid}, Int32, Ptr{Int64}, Ptr{Int64}, Ptr{Float64}, Ptr{Int64}),
                h, which, C_NULL, C_NULL, C_NULL, nnz))
    colptr = Vector{Int64}(undef, ncols + 1); rowval = Vector{Int64}(undef, nnz[]); nzval = Vector{Float64}(undef, nnz[])
    check(ccall((:pg_diffops_export_csc, libpg), Int32, (Ptr{Cvoid}, Int32, Ptr{Int64}, Ptr{Int64}, Ptr{Float64}, Ptr{Int64}),
                h, which, colptr, rowval, nzval, nnz))
    SparseMatrixCSC(nrows, ncols, colptr .+ 1, rowval .+ 1, nzval)
end
function DiffusionOps(cap::Capacity{N}) where N
    h = Ref{Ptr{Cvoid}}(C_NULL)
    check(ccall((:pg_diffops_create, libpg), Int32, (Ptr{Cvoid}, Ptr{Ptr{Cvoid}}), cap.handle, h))
    sz = cap.mesh.dims .+ 1
    M = prod(sz)
    op = DiffusionOps{N}(_export_csc(h[], 0, N * M, M), _export_csc(h[], 1, N * M, M), _export_csc(h[], 2, N * M, N * M), cap.V, sz, h[])
    finalizer(x -> ccall((:pg_diffops_destroy, libpg), Int32, (Ptr{Cvoid},), x.handle), op)
    op
end
function ∇(op::AbstractOperators, p::Vector{Float64})
    out = Vector{Float64}(undef, length(op.size) * prod(op.size))
    check(ccall((:pg_diffops_grad, libpg), Int32, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}), op.handle, p, out)); out
end
function ∇₋(op::AbstractOperators, qω::Vector{Float64}, qγ::Vector{Float64})
    out = Vector{Float64}(undef, prod(op.size))
    check(ccall((:pg_diffops_div, libpg), Int32, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}), op.handle, qω, qγ, out)); out
end

# ---------------------------------------------------------------------------------- boundary / phase (src/boundary.jl, src/phase.jl)
abstract type AbstractBoundary end
struct Dirichlet <: AbstractBoundary; value::Union{Function, Float64}; end
struct Neumann <: AbstractBoundary; value::Union{Function, Float64}; end
struct Robin <: AbstractBoundary; α::Union{Function, Float64}; β::Union{Function, Float64}; value::Union{Function, Float64}; end
struct Periodic <: AbstractBoundary end
abstract type AbstractInterfaceBC end
struct ScalarJump <: AbstractInterfaceBC; α₁; α₂; value; end
struct FluxJump <: AbstractInterfaceBC; β₁; β₂; value; end
struct BorderConditions; borders::Dict{Symbol, AbstractBoundary}; end
struct InterfaceConditions; scalar; flux; end
struct Phase
    capacity::AbstractCapacity
    operator::AbstractOperators
    source::Function
    Diffusion_coeff::Function
end

# ---------------------------------------------------------------------------------- Solver (src/solver.jl:33-42)
struct pg_bc_desc; kind::Int32; alpha::Float64; beta::Float64; value::Float64; value_array::Ptr{Float64}; end
struct pg_border_desc; key::Int32; kind::Int32; value::Float64; end
struct pg_krylov_opts; method::Int32; reltol::Float64; abstol::Float64; maxiter::Int32; check_every::Int32; warm_start::Int32; restart::Int32; end
# method kwarg -> PG_METHOD_*: :cg / IterativeSolvers.cg -> 1, :gmres / IterativeSolvers.gmres -> 2 (restarted GMRES on the device),
# everything else (`\`, bicgstabl, nothing) -> 0 = BiCGStab run to reltol
function _method_id(m)
    n = m isa Symbol ? String(m) : (m isa Function ? String(nameof(m)) : "")
    n == "cg" ? Int32(1) : (n == "gmres" ? Int32(2) : Int32(0))
end
mutable struct pg_step_info; iters::Int32; converged::Int32; resnorm::Float64; bnorm::Float64; extremum::Float64; time::Float64
    pg_step_info() = new(0, 0, 0.0, 0.0, 0.0, 0.0); end
const KEYS = Dict(:left => 0, :right => 1, :bottom => 2, :top => 3, :backward => 4, :forward => 5)

mutable struct Solver
    time_type; phase_type; equation_type
    A; b
    x::Union{Vector{Float64}, Nothing}
    ch::Vector{Any}
    states::Vector{Any}
    handle::Ptr{Cvoid}
    nunk::Int
end

coords3(c) = length(c) == 1 ? (c[1], 0.0, 0.0) : length(c) == 2 ? (c[1], c[2], 0.0) : (c[1], c[2], c[3])   # src/solver.jl:230-248
evalf(f, C, t) = [try f(coords3(c)..., t) catch e; e isa MethodError ? f(coords3(c)...) : rethrow() end for c in C]

function DiffusionUnsteadyMono(phase::Phase, bc_b::BorderConditions, bc_i::AbstractBoundary, Δt::Float64, Tᵢ::Vector{Float64}, scheme::String)
    println("Solver creation:"); println("- Monophasic problem"); println("- Unsteady problem"); println("- Diffusion problem")
    cap = phase.capacity
    kind = bc_i isa Dirichlet ? 1 : bc_i isa Neumann ? 2 : 3
    α, β = bc_i isa Robin ? (Float64(bc_i.α), Float64(bc_i.β)) : (0.0, 0.0)
    g = bc_i.value isa Function ? evalf(bc_i.value, cap.C_γ, Δt) : Float64[]
    D = [phase.Diffusion_coeff(coords3(c)...) for c in cap.C_ω]
    f = evalf(phase.source, cap.C_ω, Δt)
    borders = [pg_border_desc(KEYS[k], v isa Dirichlet ? 1 : v isa Periodic ? 4 : v isa Neumann ? 2 : 3,
                              v isa Periodic || v.value isa Function ? 0.0 : Float64(v.value)) for (k, v) in bc_b.borders if haskey(KEYS, k)]
    h = Ref{Ptr{Cvoid}}(C_NULL)
    GC.@preserve g D f Tᵢ borders begin
        desc = Ref(pg_bc_desc(kind, α, β, bc_i.value isa Function ? 0.0 : Float64(bc_i.value), isempty(g) ? C_NULL : pointer(g)))
        check(ccall((:pg_solver_create_unsteady_mono, libpg), Int32,
                    (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{pg_bc_desc}, Ptr{pg_border_desc}, Int32, Ptr{Float64}, Ptr{Float64}, Float64, Ptr{Float64}, Int32, Ptr{Ptr{Cvoid}}),
                    cap.handle, phase.operator.handle, desc, borders, length(borders), D, f, Δt, Tᵢ, scheme == "CN" ? 1 : 0, h))
    end
    s = Solver(:Unsteady, :Monophasic, :Diffusion, nothing, nothing, nothing, [], [], h[], length(Tᵢ))
    finalizer(x -> ccall((:pg_solver_destroy, libpg), Int32, (Ptr{Cvoid},), x.handle), s)
    s
end

function _state(s::Solver)
    x = zeros(s.nunk)
    check(ccall((:pg_solver_get_state, libpg), Int32, (Ptr{Cvoid}, Int64, Ptr{Float64}, Int64), s.handle, -1, x, s.nunk)); x
end

function solve_DiffusionUnsteadyMono!(s::Solver, phase::Phase, Δt::Float64, Tₑ, bc_b::BorderConditions, bc::AbstractBoundary, scheme::String;
                                      method=nothing, algorithm=nothing, kwargs...)
    s.handle == C_NULL && error("Solver is not initialized. Call a solver constructor first.")
    kw = (; kwargs...)
    opts = Ref(pg_krylov_opts(_method_id(method), get(kw, :reltol, 1e-12), get(kw, :abstol, 0.0), get(kw, :maxiter, 0), 4, get(kw, :warm_start, true) ? 1 : 0, get(kw, :restart, 0)))
    info = pg_step_info()
    t = 0.0
    check(ccall((:pg_solver_initial_solve, libpg), Int32, (Ptr{Cvoid}, Ptr{pg_krylov_opts}, Ref{pg_step_info}), s.handle, opts, info))
    s.x = _state(s); push!(s.states, s.x)
    println("Time: ", t); println("Solver Extremum: ", info.extremum)
    cap = phase.capacity
    while t < Tₑ
        t += Δt
        println("Time: ", t)
        fn, fn1 = evalf(phase.source, cap.C_ω, t), evalf(phase.source, cap.C_ω, t + Δt)
        check(ccall((:pg_solver_set_source, libpg), Int32, (Ptr{Cvoid}, Int32, Ptr{Float64}, Ptr{Float64}), s.handle, 0, fn, fn1))
        if bc.value isa Function
            gn, gn1 = evalf(bc.value, cap.C_γ, t), evalf(bc.value, cap.C_γ, t + Δt)
            check(ccall((:pg_solver_set_interface_value, libpg), Int32, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}), s.handle, gn, gn1))
        end
        check(ccall((:pg_solver_step, libpg), Int32, (Ptr{Cvoid}, Int32, Ptr{pg_krylov_opts}, Ref{pg_step_info}), s.handle, scheme == "CN" ? 1 : 0, opts, info))
        s.x = _state(s); push!(s.states, s.x)
        println("Solver Extremum: ", info.extremum)
    end
end

# ---- steady diffusion (src/solver/diffusion.jl:14-71): same blocks without V and Δt ---------------------------------
evalf0(f, C) = [f(coords3(c)...) for c in C]                                  # build_source / build_g_g without t

function DiffusionSteadyMono(phase::Phase, bc_b::BorderConditions, bc_i::AbstractBoundary)
    println("Solver creation:"); println("- Monophasic problem"); println("- Steady problem"); println("- Diffusion problem")
    cap = phase.capacity
    kind = bc_i isa Dirichlet ? 1 : bc_i isa Neumann ? 2 : 3
    α, β = bc_i isa Robin ? (Float64(bc_i.α), Float64(bc_i.β)) : (0.0, 0.0)
    g = bc_i.value isa Function ? evalf0(bc_i.value, cap.C_γ) : Float64[]
    D = [phase.Diffusion_coeff(coords3(c)...) for c in cap.C_ω]
    f = evalf0(phase.source, cap.C_ω)
    borders = [pg_border_desc(KEYS[k], v isa Dirichlet ? 1 : v isa Periodic ? 4 : v isa Neumann ? 2 : 3,
                              v isa Periodic || v.value isa Function ? 0.0 : Float64(v.value)) for (k, v) in bc_b.borders if haskey(KEYS, k)]
    h = Ref{Ptr{Cvoid}}(C_NULL)
    GC.@preserve g D f borders begin
        desc = Ref(pg_bc_desc(kind, α, β, bc_i.value isa Function ? 0.0 : Float64(bc_i.value), isempty(g) ? C_NULL : pointer(g)))
        check(ccall((:pg_solver_create_steady_mono, libpg), Int32,
                    (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{pg_bc_desc}, Ptr{pg_border_desc}, Int32, Ptr{Float64}, Ptr{Float64}, Ptr{Ptr{Cvoid}}),
                    cap.handle, phase.operator.handle, desc, borders, length(borders), D, f, h))
    end
    s = Solver(:Steady, :Monophasic, :Diffusion, nothing, nothing, nothing, [], [], h[], 2 * length(cap.C_ω))
    finalizer(x -> ccall((:pg_solver_destroy, libpg), Int32, (Ptr{Cvoid},), x.handle), s)
    s
end

function solve_DiffusionSteadyMono!(s::Solver; method=nothing, algorithm=nothing, kwargs...)
    s.handle == C_NULL && error("Solver is not initialized. Call a solver constructor first.")
    println("Solving the system:"); println("- Monophasic problem"); println("- Steady problem"); println("- Diffusion problem")
    kw = (; kwargs...)
    opts = Ref(pg_krylov_opts(_method_id(method), get(kw, :reltol, 1e-12), get(kw, :abstol, 0.0), get(kw, :maxiter, 0), 4, 0, get(kw, :restart, 0)))
    info = pg_step_info()
    check(ccall((:pg_solver_initial_solve, libpg), Int32, (Ptr{Cvoid}, Ptr{pg_krylov_opts}, Ref{pg_step_info}), s.handle, opts, info))
    s.x = _state(s)
end
# DiffusionSteadyDiph / solve_DiffusionSteadyDiph! (:88-175) bind pg_solver_create_steady_diph the same way (two
# capacities / operators, pg_jump_desc as in DiffusionUnsteadyDiph).

# ---- ConvectionOps (src/operators.jl:194-210) and the advection-diffusion drivers ---------------------------------------
# struct ConvectionOps{N}: same handle type as DiffusionOps plus the velocity; C[d] / K[d] are exported on demand with
# pg_diffops_export_csc(handle, 3 + (d-1) | 6 + (d-1), ...).  A solver constructor that receives such an operator
# assembles the advection-diffusion blocks (advectiondiffusion.jl:29-44, 95-126, 180-213).
function ConvectionOps(capacity::Capacity{N}, uₒ::NTuple{N,Vector{Float64}}, uᵧ::Vector{Float64}) where N
    op = DiffusionOps(capacity)
    GC.@preserve uₒ uᵧ begin
        ptrs = [pointer(u) for u in uₒ]
        check(ccall((:pg_diffops_set_velocity, libpg), Int32, (Ptr{Cvoid}, Ptr{Ptr{Float64}}, Ptr{Float64}), op.handle, ptrs, uᵧ))
    end
    op
end
# AdvectionDiffusionSteadyMono / SteadyDiph / UnsteadyMono / UnsteadyDiph ("BE" only: the reference's "CN" right-hand
# side for the diphasic driver omits the diffusion term, advectiondiffusion.jl:375-377) and their solve_...! functions are
# the diffusion constructors and loops above called with such an operator (DarcyFlow / DarcyFlowUnsteady likewise alias the diffusion drivers).

end # module
