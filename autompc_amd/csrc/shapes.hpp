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
//
// A shape that is NOT listed gets the same kernels at run time: ampc_set_mlp starts an
// out-of-process hipcc build of a "shape plugin" -- launch_{mppi,mlp,ilqr}.cpp + jit_plugin.cpp
// compiled with -DAMPC_JIT_PLUGIN and the registry below replaced by the one staged shape
// (-DAMPC_JIT_NX=.. etc.) -- cached on disk by shape + source hash, dlopen'ed by the next plan built
// on that model (jit_host.hpp); until it is ready the DynShape kernels run.
#pragma once
#ifdef AMPC_JIT_PLUGIN
#define AMPC_STATIC_SHAPES(X) X(0, AMPC_JIT_NX, AMPC_JIT_NU, AMPC_JIT_NO, AMPC_JIT_NH, AMPC_JIT_HPAD)
#else
#define AMPC_STATIC_SHAPES(X) \
  X(0, 17, 6, 17, 2, 256)     \
  X(1, 2, 1, 2, 2, 64)        \
  X(2, 4, 1, 4, 2, 64)        \
  X(3, 17, 6, 17, 2, 128)
#endif
