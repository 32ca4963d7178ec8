// shapes.hpp -- registry of model shapes that get a shape-specialised rollout kernel.
//
// X(id, nx, nu, obs_dim, n_hidden, hpad): state / control / observation dimensions, hidden layers,
// padded hidden width (64 -> W = 4, NT = 1; 128 -> W = 8, NT = 1; 192 -> W = 4, NT = 3; 256 -> W = 8,
// NT = 2).  A staged model whose dimensions match an entry runs mppi_rollout_kernel<..., StaticShape>
// (16- and 32-row tiles, any activation, any cost); everything else runs the DynShape instantiation
// of the same kernel.  The entries are the systems of the reference's benchmark suite with the
// reference's default network (2 hidden layers; mlp.py:112-135), i.e. BASELINE.json configs 2-5:
//   0  HalfCheetah  17 states, 6 controls, 2 x 256    (benchmarks/halfcheetah.py)
//   1  Pendulum      2 states, 1 control,  2 x 64     (benchmarks/pendulum.py)
//   2  CartPole      4 states, 1 control,  2 x 64     (benchmarks/cartpole.py)
//   3  HalfCheetah with a 2 x 128 network (the config-space default hidden size, mlp.py:120-124)
// Adding a shape = adding a line (costs ~1 minute of build time per precision).
#pragma once
#define AMPC_STATIC_SHAPES(X) \
  X(0, 17, 6, 17, 2, 256)     \
  X(1, 2, 1, 2, 2, 64)        \
  X(2, 4, 1, 4, 2, 64)        \
  X(3, 17, 6, 17, 2, 128)
