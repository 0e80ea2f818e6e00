# LudwigHIP.jl - the reference-side binding of libludwig_hip.so (include/ludwig_hip.h).
#
# Drop this file next to the reference's src/ and `include("LudwigHIP.jl")` after blocks.jl. It replaces, for an
# MI355X, the four device-facing calls of the hot path:
#     adapt(backend, level)                 (src/main.jl:98, src/blocks.jl:67-87)      -> LudwigHIP.adapt_level
#     perform_timestep_v2!(level, ...)      (src/physics_v2.jl:26-97)                   -> LudwigHIP.perform_timestep!
#     copy_to_old!(level, f_in, vel_in)     (src/blocks.jl:199-205)                     -> LudwigHIP.copy_to_old!
#     KernelAbstractions.synchronize        (src/solver_control.jl:164)                 -> LudwigHIP.synchronize
# NOT TESTED HERE: the build image has no Julia. It is a thin ccall layer; every struct mirrors include/ludwig_hip.h.
module LudwigHIP

const LIB = get(ENV, "LUDWIG_HIP_LIB", joinpath(@__DIR__, "..", "open_ludwig_amd", "csrc", "libludwig_hip.so"))

# enum LudwigField
const F, F_TEMP, F_POST, F_OLD, RHO, RHO_OLD, VEL, VEL_TEMP, VEL_OLD, OBSTACLE, SPONGE, WALL_DIST = Int32.(0:11)

struct LevelHost               # LudwigLevelHost
    level_id::Int32; n_blocks::Int32; n_owned::Int32; tau::Float32
    grid_dim_x::Int32; grid_dim_y::Int32; grid_dim_z::Int32
    block_pointer::Ptr{Int32}; neighbor_table::Ptr{Int32}
    map_x::Ptr{Int32}; map_y::Ptr{Int32}; map_z::Ptr{Int32}
    obstacle::Ptr{UInt8}; sponge::Ptr{Float32}; wall_dist::Ptr{Float32}
    enable_temporal_interpolation::Int32; n_boundary_cells::Int32
    bouzidi_q_map::Ptr{UInt16}; bouzidi_cell_block::Ptr{Int32}
    bouzidi_cell_x::Ptr{Int8}; bouzidi_cell_y::Ptr{Int8}; bouzidi_cell_z::Ptr{Int8}
    comm_boundary::Ptr{UInt8}
    store_post_collision_everywhere::Int32   # 0: f_post_collision only where the Bouzidi kernel reads it
end

struct StepFlags               # LudwigStepFlags
    domain_nx::Int32; domain_ny::Int32; domain_nz::Int32
    is_symmetric::Int32; wall_model_active::Int32; use_temporal_interp::Int32; sponge_blend_distributions::Int32
    c_wale::Float32; nu_sgs_background::Float32; inlet_turbulence::Float32; q_min_threshold::Float32
end

mutable struct DeviceLevel
    handle::Ptr{Cvoid}
    level_id::Int; tau::Float32; n_blocks::Int
    has_temporal::Bool; bouzidi_enabled::Bool; n_boundary_cells::Int
end

last_error() = unsafe_string(ccall((:ludwig_last_error, LIB), Cstring, ()))
check(rc::Cint) = rc == 0 ? nothing : error("libludwig_hip error $rc: $(last_error())")

"""adapt(backend, level) for an MI355X: uploads every array of a host BlockLevel (src/blocks.jl:16-65)."""
function adapt_level(level, device::Integer = 0)
    n = length(level.active_block_coords)
    obstacle_u8 = Array{UInt8}(level.obstacle)
    GC.@preserve level obstacle_u8 begin
        h = LevelHost(Int32(level.level_id), Int32(n), Int32(0), level.tau,
                      Int32(size(level.block_pointer, 1)), Int32(size(level.block_pointer, 2)), Int32(size(level.block_pointer, 3)),
                      pointer(level.block_pointer), pointer(level.neighbor_table),
                      pointer(level.map_x), pointer(level.map_y), pointer(level.map_z),
                      pointer(obstacle_u8), pointer(level.sponge), pointer(level.wall_dist),
                      Int32(length(level.f_old) > 27), Int32(level.n_boundary_cells),
                      level.bouzidi_enabled ? Ptr{UInt16}(pointer(level.bouzidi_q_map)) : Ptr{UInt16}(C_NULL),
                      level.bouzidi_enabled ? pointer(level.bouzidi_cell_block) : Ptr{Int32}(C_NULL),
                      level.bouzidi_enabled ? pointer(level.bouzidi_cell_x) : Ptr{Int8}(C_NULL),
                      level.bouzidi_enabled ? pointer(level.bouzidi_cell_y) : Ptr{Int8}(C_NULL),
                      level.bouzidi_enabled ? pointer(level.bouzidi_cell_z) : Ptr{Int8}(C_NULL),
                      Ptr{UInt8}(C_NULL), Int32(0))
        out = Ref{Ptr{Cvoid}}(C_NULL)
        check(ccall((:ludwig_level_create, LIB), Cint, (Ref{LevelHost}, Cint, Ref{Ptr{Cvoid}}), h, device, out))
        d = DeviceLevel(out[], level.level_id, level.tau, n, length(level.f_old) > 27, level.bouzidi_enabled, level.n_boundary_cells)
        for (fld, arr) in ((F, level.f), (F_TEMP, level.f_temp), (RHO, level.rho), (VEL, level.vel), (VEL_TEMP, level.vel_temp))
            upload!(d, fld, arr)
        end
        finalizer(x -> ccall((:ludwig_level_destroy, LIB), Cvoid, (Ptr{Cvoid},), x.handle), d)
        return d
    end
end

upload!(d::DeviceLevel, field::Int32, a::Array) =
    GC.@preserve a check(ccall((:ludwig_level_upload, LIB), Cint, (Ptr{Cvoid}, Cint, Ptr{Cvoid}, Csize_t), d.handle, field, pointer(a), sizeof(a)))

"""Array(level.field): download into a preallocated host array of the reference's shape."""
download!(a::Array, d::DeviceLevel, field::Int32) =
    GC.@preserve a check(ccall((:ludwig_level_download, LIB), Cint, (Ptr{Cvoid}, Cint, Ptr{Cvoid}, Csize_t), d.handle, field, pointer(a), sizeof(a)))

"""init_eq! (src/main.jl:109-134)"""
init_equilibrium!(d::DeviceLevel) = check(ccall((:ludwig_init_equilibrium, LIB), Cint, (Ptr{Cvoid},), d.handle))

"""
perform_timestep_v2! (src/physics_v2.jl:26-97). The reference passes f_out/f_in/vel_out/vel_in explicitly; its only callers
(src/solver_control.jl:35-41) derive them from the parity of `timestep`, which is what the library does. `parent === nothing`
is level 1, exactly like `parent_f === nothing`.
"""
function perform_timestep!(d::DeviceLevel, parent::Union{DeviceLevel,Nothing}, parent_tau::Float32, u_curr::Float32,
                           flags::StepFlags, timestep::Integer, temporal_weight::Float32)
    p = parent === nothing ? C_NULL : parent.handle
    check(ccall((:ludwig_step, LIB), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Int64, Cfloat, Cfloat, Cfloat, Ref{StepFlags}),
                d.handle, p, Int64(timestep), u_curr, parent_tau, temporal_weight, flags))
end

"""copy_to_old!(level, f_in, vel_in) for the step `timestep` about to run (src/blocks.jl:199-205)."""
copy_to_old!(d::DeviceLevel, timestep::Integer) = check(ccall((:ludwig_save_old, LIB), Cint, (Ptr{Cvoid}, Int64), d.handle, Int64(timestep)))

synchronize(d::DeviceLevel) = check(ccall((:ludwig_sync, LIB), Cint, (Ptr{Cvoid},), d.handle))

"""compute_flow_stats(level).rho_min (src/diagnostics.jl:56-94), reduced on the device."""
function rho_min(d::DeviceLevel)
    v = Ref{Cfloat}(0)
    check(ccall((:ludwig_level_rho_min, LIB), Cint, (Ptr{Cvoid}, Ref{Cfloat}), d.handle, v))
    return v[]
end

"""
Multi-GPU hosts only: a HIP stream for the stepping kernels that leaves `reserved_cus` compute units to the halo exchange
(`ludwig_stream_create`, include/ludwig_hip.h); hand it to `ludwig_level_set_stream`. No counterpart in the reference.
"""
function stream_create(device::Integer, reserved_cus::Integer)
    s = Ref{Ptr{Cvoid}}(C_NULL)
    check(ccall((:ludwig_stream_create, LIB), Cint, (Cint, Cint, Ref{Ptr{Cvoid}}), Cint(device), Cint(reserved_cus), s))
    return s[]
end
stream_destroy(device::Integer, s::Ptr{Cvoid}) = check(ccall((:ludwig_stream_destroy, LIB), Cint, (Cint, Ptr{Cvoid}), Cint(device), s))

# ---- multi-GPU (no counterpart in the reference: single device, src/main.jl:75). One Julia process per GPU; the host carries the
# 128-byte RCCL id from rank 0 to the others by whatever it already has (MPI.jl: MPI.Bcast!, a shared file, Sockets). ----
struct Comm
    handle::Ptr{Cvoid}
end
"""ncclGetUniqueId through the library: call on ONE rank, broadcast the 128 bytes."""
function comm_unique_id()
    id = Vector{UInt8}(undef, 128)
    GC.@preserve id check(ccall((:ludwig_comm_unique_id, LIB), Cint, (Ptr{Cvoid},), pointer(id)))
    return id
end
function Comm(id::Vector{UInt8}, rank::Integer, world::Integer, device::Integer)
    out = Ref{Ptr{Cvoid}}(C_NULL)
    GC.@preserve id check(ccall((:ludwig_comm_create, LIB), Cint, (Ptr{Cvoid}, Cint, Cint, Cint, Ref{Ptr{Cvoid}}), pointer(id), Cint(rank), Cint(world), Cint(device), out))
    return Comm(out[])
end
destroy!(c::Comm) = ccall((:ludwig_comm_destroy, LIB), Cvoid, (Ptr{Cvoid},), c.handle)
"""small levels: queue the exchange on the level's own stream (no overlap, no cross-stream hand-over; `wait!` becomes a no-op)"""
in_stream!(p, on::Bool) = check(ccall((:ludwig_halo_plan_in_stream, LIB), Cint, (Ptr{Cvoid}, Cint), p.handle, Cint(on)))
"""
Bouzidi level cut over ranks: the f_post_collision elements a PEER's links read across the cut (= this rank's group-2 send list,
0-based element offsets in the [8,8,8,n_blocks,27] layout). The step then stores the rows with a reader instead of every block.
"""
function add_post_collision_readers!(d::DeviceLevel, offsets::Vector{Int64})
    GC.@preserve offsets check(ccall((:ludwig_level_add_post_collision_readers, LIB), Cint, (Ptr{Cvoid}, Ptr{Int64}, Int64),
                                     d.handle, isempty(offsets) ? Ptr{Int64}(C_NULL) : pointer(offsets), Int64(length(offsets))))
end
"""in-place all-reduce of a few Float32 diagnostics (op 0 sum, 2 max, 3 min): rho_min, the nine force sums"""
allreduce!(c::Comm, v::Vector{Float32}, op::Integer) =
    GC.@preserve v check(ccall((:ludwig_comm_allreduce_f32, LIB), Cint, (Ptr{Cvoid}, Ptr{Cfloat}, Int32, Int32), c.handle, pointer(v), Int32(length(v)), Int32(op)))

"""LudwigHaloPlanDesc (include/ludwig_hip.h): per group g = 1..4 (populations, velocity, f_post_collision, rho) the element offsets
this rank sends / receives, all peers concatenated; offsets are 0-based positions in the level's arrays as the reference lays them out."""
struct HaloPlanDesc
    n_peers::Int32
    peer_ranks::Ptr{Int32}
    send_count::NTuple{4,Ptr{Int64}}
    recv_count::NTuple{4,Ptr{Int64}}
    send_index::NTuple{4,Ptr{Int64}}
    recv_index::NTuple{4,Ptr{Int64}}
end
struct HaloPlan
    handle::Ptr{Cvoid}
end
function HaloPlan(d::DeviceLevel, c::Union{Comm,Nothing}, peers::Vector{Int32}, send_count::NTuple{4,Vector{Int64}}, recv_count::NTuple{4,Vector{Int64}},
                  send_index::NTuple{4,Vector{Int64}}, recv_index::NTuple{4,Vector{Int64}})
    out = Ref{Ptr{Cvoid}}(C_NULL)
    GC.@preserve peers send_count recv_count send_index recv_index begin
        desc = HaloPlanDesc(Int32(length(peers)), pointer(peers), map(pointer, send_count), map(pointer, recv_count), map(pointer, send_index), map(pointer, recv_index))
        check(ccall((:ludwig_halo_plan_create, LIB), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Ref{HaloPlanDesc}, Ref{Ptr{Cvoid}}),
                    d.handle, c === nothing ? C_NULL : c.handle, desc, out))
    end
    return HaloPlan(out[])
end
destroy!(p::HaloPlan) = ccall((:ludwig_halo_plan_destroy, LIB), Cvoid, (Ptr{Cvoid},), p.handle)
"""one exchange, queued behind what the level's stream holds: group groups[i] (0-based) moves field fields[i]"""
exchange!(p::HaloPlan, groups::Vector{Int32}, fields::Vector{Int32}) =
    GC.@preserve groups fields check(ccall((:ludwig_halo_exchange, LIB), Cint, (Ptr{Cvoid}, Int32, Ptr{Int32}, Ptr{Int32}), p.handle, Int32(length(groups)), pointer(groups), pointer(fields)))
wait!(p::HaloPlan) = check(ccall((:ludwig_halo_wait, LIB), Cint, (Ptr{Cvoid},), p.handle))

"""
perform_timestep_v2! of a level spread over ranks: interior blocks, wait for the previous exchange, boundary blocks, [f_post halo,
Bouzidi], this step's exchange left in flight (ludwig_step_distributed). Drop-in for `perform_timestep!` in `recursive_step!` below
when the level has a plan.
"""
function perform_timestep_distributed!(d::DeviceLevel, plan::HaloPlan, parent::Union{DeviceLevel,Nothing}, parent_tau::Float32, u_curr::Float32,
                                       flags::StepFlags, timestep::Integer, temporal_weight::Float32)
    p = parent === nothing ? C_NULL : parent.handle
    check(ccall((:ludwig_step_distributed, LIB), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Int64, Cfloat, Cfloat, Cfloat, Ref{StepFlags}),
                d.handle, plan.handle, p, Int64(timestep), u_curr, parent_tau, temporal_weight, flags))
end

"""
recursive_step! with the device calls swapped (src/solver_control.jl:21-143): same order, same parity, same weights.
"""
function recursive_step!(grids::Vector{DeviceLevel}, lvl::Int, t_sub::Int, parent, parent_tau::Float32, tw::Float32,
                         u_vel::Float32, flags::StepFlags)
    lvl > length(grids) && return
    level = grids[lvl]
    has_children = lvl < length(grids)
    if has_children && flags.use_temporal_interp == 1 && level.has_temporal
        copy_to_old!(level, t_sub)
    end
    perform_timestep!(level, parent, parent_tau, u_vel, flags, t_sub, tw)
    if has_children
        recursive_step!(grids, lvl + 1, 2 * t_sub, level, level.tau, 0.0f0, u_vel, flags)
        recursive_step!(grids, lvl + 1, 2 * t_sub + 1, level, level.tau, 0.5f0, u_vel, flags)
    end
end

"""execute_timestep_batch! (src/solver_control.jl:145-165): the whole batch in one ccall (the library runs the same
recursion); `recursive_step!` above is the call-by-call equivalent."""
function execute_timestep_batch!(grids::Vector{DeviceLevel}, t_start::Int, batch_size::Int, u_curr::Float32, flags::StepFlags)
    handles = Ptr{Cvoid}[g.handle for g in grids]
    GC.@preserve handles check(ccall((:ludwig_execute_timestep_batch, LIB), Cint,
                                     (Ptr{Ptr{Cvoid}}, Int32, Int64, Int32, Cfloat, Ref{StepFlags}),
                                     handles, Int32(length(grids)), Int64(t_start), Int32(batch_size), u_curr, flags))
end

end # module
