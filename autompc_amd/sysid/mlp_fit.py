"""Fitting MLP surrogates with PyTorch-ROCm: the reference's training loop, one model at a time or K at once.

What it replaces.  ``MLP.train`` (reference autompc/sysid/mlp.py:177-217): normalise ``[x, u]`` and
``dy = x' - x`` column-wise, then ``n_train_iters`` epochs of Adam (``lr``) on SmoothL1 over shuffled
mini-batches of ``n_batch`` rows (``DataLoader(shuffle=True)``, the ragged last batch kept).  The
tuner fits one such model per configuration (``pipeline.py:138-145`` inside ``eval_cfg``,
``pipeline_tuner.py:213-215``), which is what bounds a model-axis search once the closed-loop evaluation
runs at device speed.

Training stays PyTorch; what this module adds is the SHAPE of the torch program:

* ``fit_mlps`` trains K models **in lockstep**: parameters stacked ``[K, out, in + 1]`` (the bias is the weight
  of a constant-1 input column), one ``bmm`` per layer for all K, a backward pass written out with the kernels
  autograd itself runs (``smooth_l1_loss_backward``, ``threshold_backward`` / ``tanh_backward`` / ...) -- one
  ``bmm`` per layer yields the gradients of weights and bias together --, and torch's fused Adam kernel over ONE
  flat buffer holding every parameter of every model (per-model learning rates applied as a per-element
  vector).  Models of one depth and activation but different widths are zero-padded to the group's widest layer
  (a mask keeps the padding at zero).  A step is 18 small kernels whatever K is, so K models cost about what one
  costs.
* on a GPU the steps are captured into **HIP graphs** (``torch.cuda.CUDAGraph``: a chunk of up to 64
  optimiser steps per graph, the mini-batches of a chunk gathered by one indexing kernel into a static
  buffer), which removes the per-kernel launch cost that dominates a 64-row step.
* every model keeps the random streams the reference gives it: ``torch.manual_seed(seed)`` before the
  layers are built (``mlp.py:160-161``), ``torch.manual_seed(seed)`` again at the top of ``train`` and
  the two draws per epoch a ``DataLoader`` + ``RandomSampler`` take from the global generator
  (torch/utils/data/dataloader.py ``_base_seed``, sampler.py ``RandomSampler.__iter__``) -- per-model
  ``torch.Generator`` objects here, so K lockstep models see exactly the initial weights and mini-batch
  order K sequential reference fits would (``tests/golden/mlpfit_*.npz`` pins that against the reference).
* the fitted parameters stay on the device; ``MLP.stage_into`` hands their addresses to
  ``ampc_set_mlp_dev`` (no host round trip).  The numpy copies ``get_parameters`` needs are fetched lazily.

``fit_reference_style`` is the plain ``nn.Linear`` + ``torch.optim.Adam`` + autograd loop, one model, kept
as the cross-check of the hand-written backward pass / update and as the sequential baseline of
``tools/model_axis_rate.py``.
"""
import math
import time

import numpy as np

ACTS = ("relu", "tanh", "sigmoid", "selu")
_BETA1, _BETA2, _EPS = 0.9, 0.999, 1e-8        # torch.optim.Adam defaults (mlp.py:197 passes lr only)
_SELU_ALPHA, _SELU_SCALE = 1.6732632423543772848170429916717, 1.0507009873554804934193349852946
CHUNK = 64                                     # optimiser steps per captured graph


def training_arrays(trajs):
    """(XU, dY, xu_means, xu_std, dy_means, dy_std) as mlp.py:179-190 builds them."""
    X = np.concatenate([t.obs[:-1, :] for t in trajs])
    dY = np.concatenate([t.obs[1:, :] - t.obs[:-1, :] for t in trajs])
    U = np.concatenate([t.ctrls[:-1, :] for t in trajs])
    XU = np.concatenate([X, U], axis=1)
    return XU, dY, XU.mean(axis=0), XU.std(axis=0), dY.mean(axis=0), dY.std(axis=0)


def normalised(XU, dY, xu_means, xu_std, dy_means, dy_std):
    """The feed / target arrays of the fit (transform_input, mlp.py:20-24: column by column)."""
    return (XU - xu_means) / xu_std, (dY - dy_means) / dy_std


# -- the reference's random streams -------------------------------------------------------------------
def seeded_generator(seed):
    import torch
    g = torch.Generator()
    g.manual_seed(int(seed))
    return g


def linear_init(gen, fan_in, fan_out):
    """torch.nn.Linear.reset_parameters from generator `gen`: float32 draws (the reference builds the net in
    float32 and then calls .double(), mlp.py:161-165), kaiming_uniform(a=sqrt(5)) weight, then the bias."""
    import torch
    gain = math.sqrt(2.0 / (1 + math.sqrt(5.0) ** 2))
    bound_w = math.sqrt(3.0) * (gain / math.sqrt(fan_in))
    w = torch.empty(fan_out, fan_in, dtype=torch.float32).uniform_(-bound_w, bound_w, generator=gen)
    bound_b = 1 / math.sqrt(fan_in) if fan_in > 0 else 0
    b = torch.empty(fan_out, dtype=torch.float32).uniform_(-bound_b, bound_b, generator=gen)
    return w.double(), b.double()


def initial_parameters(seed, dims):
    """Weights / biases of ForwardNet(dims) built right after torch.manual_seed(seed) (mlp.py:36-42: hidden
    layers in order, then the output layer)."""
    gen = seeded_generator(seed)
    ws, bs = [], []
    for a, b in zip(dims[:-1], dims[1:]):
        w, bias = linear_init(gen, a, b)
        ws.append(w)
        bs.append(bias)
    return ws, bs


def epoch_order(gen, n):
    """The row order of one epoch: what iterating DataLoader(shuffle=True) draws from the global generator
    `gen` stands for -- the iterator's base seed (discarded here, it seeds workers), the sampler's seed, and
    a randperm from a fresh generator seeded with that."""
    import torch
    torch.empty((), dtype=torch.int64).random_(generator=gen)
    seed = int(torch.empty((), dtype=torch.int64).random_(generator=gen).item())
    g = torch.Generator()
    g.manual_seed(seed)
    return torch.randperm(n, generator=g)


def _act(name):
    import torch
    return {"relu": torch.relu, "tanh": torch.tanh, "sigmoid": torch.sigmoid, "selu": torch.selu}[name]


# -- one model, the reference's own program ----------------------------------------------------------------
def fit_reference_style(dims, act, feed, target, n_iter, n_batch, lr, init_seed, train_seed=100, device=None):
    """nn.Linear layers + torch.optim.Adam + SmoothL1Loss + autograd, mini-batches in the reference's order;
    feed / target are float64 tensors (anywhere), the data is moved to `device` once.  Returns the weights and
    biases as lists of float64 tensors on `device`."""
    import torch
    device = torch.device(device or "cpu")
    torch.manual_seed(int(init_seed))
    layers = []
    for a, b in zip(dims[:-2], dims[1:-1]):
        layers += [torch.nn.Linear(a, b), {"relu": torch.nn.ReLU, "tanh": torch.nn.Tanh,
                                           "sigmoid": torch.nn.Sigmoid, "selu": torch.nn.SELU}[act]()]
    layers.append(torch.nn.Linear(dims[-2], dims[-1]))
    net = torch.nn.Sequential(*layers).double().to(device)
    gen = seeded_generator(train_seed)
    feed, target = feed.to(device), target.to(device)
    opt = torch.optim.Adam(net.parameters(), lr=lr)
    loss_fn = torch.nn.SmoothL1Loss()
    n = feed.shape[0]
    for _ in range(int(n_iter)):
        perm = epoch_order(gen, n).to(device)
        for s in range(0, n, n_batch):
            idx = perm[s:s + n_batch]
            opt.zero_grad()
            loss = loss_fn(net(feed[idx]), target[idx])
            loss.backward()
            opt.step()
    lin = [m for m in net if isinstance(m, torch.nn.Linear)]
    return [m.weight.detach() for m in lin], [m.bias.detach() for m in lin]


# -- K models in lockstep ----------------------------------------------------------------------------------
class LockstepFit:
    """K MLPs of one depth and one activation over one data set; see the module docstring.

    The bias of a layer is the weight of a constant-1 input column: parameters are stacked `[K, out, in + 1]`,
    activations carry a last column of ones (written once into static buffers), so a layer is ONE `bmm` forward and
    ONE `bmm` for the gradients of weights and bias together (no bias broadcast, no bias-gradient reduction)."""

    def __init__(self, dims_list, act, lrs, init_seeds, feed, target, n_batch, train_seeds=None, device=None,
                 use_graphs=None):
        import torch
        self.torch = torch
        K = self.K = len(dims_list)
        depth = len(dims_list[0])
        if any(len(d) != depth for d in dims_list):
            raise ValueError("lockstep models must have the same number of layers")
        if any(d[0] != dims_list[0][0] or d[-1] != dims_list[0][-1] for d in dims_list):
            raise ValueError("lockstep models must share the input and output width")
        if act not in ACTS:
            raise NotImplementedError("Currently supported nonlinearity: relu, selu, tanh, sigmoid")
        self.dims_list = [tuple(int(v) for v in d) for d in dims_list]
        self.act = act
        self.nl = depth - 1                                      # linear layers
        self.dmax = [max(d[i] for d in self.dims_list) for i in range(depth)]
        self.device = torch.device(device or ("cuda" if torch.cuda.is_available() else "cpu"))
        self.on_gpu = self.device.type == "cuda"
        self.use_graphs = self.on_gpu if use_graphs is None else bool(use_graphs and self.on_gpu)
        self.n_batch = int(n_batch)
        dev, f64 = self.device, torch.float64
        feed = feed.to(dev, f64)
        self.feed = torch.cat([feed, torch.ones(feed.shape[0], 1, dtype=f64, device=dev)], dim=1).contiguous()
        self.target = target.to(dev, f64).contiguous()
        self.n = int(self.feed.shape[0])
        train_seeds = [100] * K if train_seeds is None else list(train_seeds)
        self.gens = [seeded_generator(s) for s in train_seeds]
        # ONE flat buffer for all parameters: per layer the stacked [K, out, in + 1] (last column: the bias);
        # gradients, both Adam moments, the learning rates and the padding mask alike
        sizes = [K * self.dmax[l + 1] * (self.dmax[l] + 1) for l in range(self.nl)]
        total = sum(sizes)
        self.flat = torch.zeros(total, dtype=f64, device=dev)
        self.grad = torch.zeros_like(self.flat)
        self.m, self.v = torch.zeros_like(self.flat), torch.zeros_like(self.flat)
        host_flat = torch.zeros(total, dtype=f64)
        host_mask = torch.zeros(total, dtype=f64)
        host_lr = torch.zeros(total, dtype=f64)

        def views(buf):
            out, o = [], 0
            for l in range(self.nl):
                n = sizes[l]
                out.append(buf[o:o + n].view(K, self.dmax[l + 1], self.dmax[l] + 1))
                o += n
            return out
        hw, mw, lw = views(host_flat), views(host_mask), views(host_lr)
        for k, (d, seed, lr) in enumerate(zip(self.dims_list, init_seeds, lrs)):
            ws, bs = initial_parameters(seed, d)
            for l in range(self.nl):
                hw[l][k, :d[l + 1], :d[l]] = ws[l]
                hw[l][k, :d[l + 1], self.dmax[l]] = bs[l]
                mw[l][k, :d[l + 1], :d[l]] = 1.0
                mw[l][k, :d[l + 1], self.dmax[l]] = 1.0
                lw[l][k] = float(lr)
        self.flat.copy_(host_flat)
        self.padded = bool((host_mask == 0).any())
        self.mask = host_mask.to(dev) if self.padded else None
        self.lr_flat = host_lr.to(dev)
        self.W, self.gW = views(self.flat), views(self.grad)
        self.steps_done = 0
        self.dirn = torch.zeros_like(self.flat)                     # Adam's direction at lr = 1
        # Adam's step count as the fused kernel wants it, on the device: entry j of a small table = the count at the
        # j-th step of the running chunk, written by ONE launch per chunk (not one increment per step)
        self._count = 0
        self._ar = torch.arange(1, CHUNK + 2, dtype=torch.float32, device=dev)
        self._tt = torch.zeros(CHUNK + 1, dtype=torch.float32, device=dev)
        self._k = torch.tensor(float(K), dtype=f64, device=dev)
        piece = max(1024, -(-total // 64))
        self._dirn_v, self._grad_v = list(torch.split(self.dirn, piece)), list(torch.split(self.grad, piece))
        self._m_v, self._v_v = list(torch.split(self.m, piece)), list(torch.split(self.v, piece))
        self._t_v = [[self._tt[j]] * len(self._dirn_v) for j in range(CHUNK + 1)]
        self._acts = {}                    # mini-batch rows -> the hidden layers' activation buffers (ones column set)
        self._graphs = {}
        self.kernel_s = 0.0

    def _act_buffers(self, nb):
        """Static activation buffers [K, nb, width + 1] of the hidden layers, last column = 1 (the bias input)."""
        if nb not in self._acts:
            torch = self.torch
            self._acts[nb] = [torch.ones(self.K, nb, self.dmax[l + 1] + 1, dtype=torch.float64, device=self.device)
                              for l in range(self.nl - 1)]
        return self._acts[nb]

    # .. one optimiser step on mini-batches x [K, nb, in + 1] (ones column included), y [K, nb, out] .................
    def _step(self, x, y, j):
        """The j-th step of the running chunk.  Forward, backward and update with the kernels autograd would run for the reference's program -- one
        launch each: smooth_l1_loss_backward, <activation>_backward, the fused Adam kernel -- on the stacked
        tensors."""
        torch = self.torch
        act, nl = self.act, self.nl
        aten = torch.ops.aten
        bufs = self._act_buffers(x.shape[1])
        a, z = [x], []
        for l in range(nl - 1):
            zl = torch.bmm(a[-1], self.W[l].transpose(1, 2))
            h = bufs[l][:, :, :self.dmax[l + 1]]                 # (the ones column stays)
            if act == "relu":
                torch.clamp_min(zl, 0.0, out=h)
            elif act == "tanh":
                torch.tanh(zl, out=h)
            elif act == "sigmoid":
                torch.sigmoid(zl, out=h)
            else:
                aten.elu.out(zl, _SELU_ALPHA, _SELU_SCALE, 1.0, out=h)
            z.append(zl)
            a.append(bufs[l])
        out = torch.bmm(a[-1], self.W[nl - 1].transpose(1, 2))
        # SmoothL1Loss(beta = 1), mean over EACH model's nb * out entries: the op averages over all K models'
        # entries, the incoming gradient K undoes that (for K = 1 this is the reference's own backward kernel)
        g = aten.smooth_l1_loss_backward(self._k, out, y, 1, 1.0)
        for l in range(nl - 1, -1, -1):
            torch.bmm(g.transpose(1, 2), a[l], out=self.gW[l])   # weights' and (last column) bias' gradients
            if l > 0:
                g = torch.bmm(g, self.W[l])[:, :, :self.dmax[l]]
                h = a[l][:, :, :self.dmax[l]]
                if act == "relu":
                    g = aten.threshold_backward(g, z[l - 1], 0.0)
                elif act == "tanh":
                    g = aten.tanh_backward(g, h)
                elif act == "sigmoid":
                    g = aten.sigmoid_backward(g, h)
                else:
                    g = aten.elu_backward(g, _SELU_ALPHA, _SELU_SCALE, 1.0, False, z[l - 1])
        if self.padded:
            self.grad.mul_(self.mask)
        # Adam over the flat buffer (torch's fused implementation, the step count on the device): run with lr = 1 on
        # a zeroed direction buffer, so that every model's own learning rate can scale its part of the direction
        # afterwards (one lr per fused call is all the kernel takes).  Handed over as ~64 slices: the multi-tensor
        # kernel gives every tensor chunk ONE workgroup -- the whole buffer as one tensor is ten 65536-element chunks
        # on ten workgroups, 58 us; sliced, 2 x 15.
        self.dirn.zero_()
        torch._fused_adam_(self._dirn_v, self._grad_v, self._m_v, self._v_v, [], self._t_v[j], lr=1.0, beta1=_BETA1,
                           beta2=_BETA2, weight_decay=0.0, eps=_EPS, amsgrad=False, maximize=False)
        self.flat.addcmul_(self.dirn, self.lr_flat)

    # .. a chunk of `steps` consecutive optimiser steps on rows idx [K, steps * nb] .............................
    def _run_chunk(self, idx, steps, nb):
        torch = self.torch
        K = self.K
        self._act_buffers(nb)                                  # (allocated outside any capture)
        if not self.use_graphs:
            torch.add(self._ar, float(self._count), out=self._tt)
            x = self.feed[idx].view(K, steps, nb, -1)
            y = self.target[idx].view(K, steps, nb, -1)
            for j in range(steps):
                self._step(x[:, j], y[:, j], j)
            self._count += steps
            return
        key = (steps, nb)
        if key not in self._graphs:
            sidx = torch.zeros(K, steps * nb, dtype=torch.int64, device=self.device)
            snap = (self.flat.clone(), self.m.clone(), self.v.clone())
            torch.add(self._ar, float(self._count), out=self._tt)

            def body():
                x = self.feed[sidx].view(K, steps, nb, -1)
                y = self.target[sidx].view(K, steps, nb, -1)
                for j in range(steps):
                    self._step(x[:, j], y[:, j], j)
            side = torch.cuda.Stream(device=self.device)
            side.wait_stream(torch.cuda.current_stream(self.device))
            with torch.cuda.stream(side):                      # warm-up outside the capture (library workspaces)
                x = self.feed[sidx[:, :nb]].view(K, 1, nb, -1)
                y = self.target[sidx[:, :nb]].view(K, 1, nb, -1)
                self._step(x[:, 0], y[:, 0], 0)
            torch.cuda.current_stream(self.device).wait_stream(side)
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                body()
            # warm-up and capture ran (resp. recorded) real updates: restore the state they touched
            self.flat.copy_(snap[0]); self.m.copy_(snap[1]); self.v.copy_(snap[2])
            self._graphs[key] = (graph, sidx)
        graph, sidx = self._graphs[key]
        sidx.copy_(idx)
        torch.add(self._ar, float(self._count), out=self._tt)
        graph.replay()
        self._count += steps

    def run(self, n_iter):
        """`n_iter` more epochs."""
        torch = self.torch
        n, nb, K = self.n, self.n_batch, self.K
        n_full, rag = divmod(n, nb)
        per_epoch = n_full + (1 if rag else 0)
        n_chunks = max(1, -(-n_full // CHUNK))
        t0 = time.perf_counter()
        for _ in range(int(n_iter)):
            perm = torch.stack([epoch_order(g, n) for g in self.gens]).to(self.device, non_blocking=True)
            done = 0
            for c in range(n_chunks):                                # chunks of nearly equal size: two graphs
                steps = n_full // n_chunks + (1 if c < n_full % n_chunks else 0)
                if steps:
                    self._run_chunk(perm[:, done * nb:(done + steps) * nb], steps, nb)
                done += steps
            if rag:
                self._run_chunk(perm[:, n_full * nb:], 1, rag)
            self.steps_done += per_epoch
        if self.on_gpu:
            torch.cuda.synchronize(self.device)
        self.kernel_s += time.perf_counter() - t0

    def parameters(self, k):
        """Model k's weights [out][in] and biases [out]: contiguous float64 tensors on the fit's device."""
        d = self.dims_list[k]
        ws = [self.W[l][k, :d[l + 1], :d[l]].contiguous() for l in range(self.nl)]
        bs = [self.W[l][k, :d[l + 1], self.dmax[l]].contiguous() for l in range(self.nl)]
        return ws, bs


def fit_mlps(models, trajs, train_seed=100, device=None, use_graphs=None):
    """Fit every model of `models` (autompc_amd.sysid.MLP, any mix of shapes) on `trajs`, each exactly as its
    own ``train(trajs, seed=train_seed)`` would, grouped into lockstep fits by (depth, activation, epochs,
    batch size).  Returns {"groups": n, "fit_s": wall seconds, "steps": optimiser steps per model}."""
    import torch
    t0 = time.perf_counter()
    XU, dY, xm, xs, dm, ds = training_arrays(trajs)
    feed_np, target_np = normalised(XU, dY, xm, xs, dm, ds)
    feed, target = torch.from_numpy(feed_np), torch.from_numpy(target_np)
    groups = {}
    for m in models:
        n_iter, n_batch, _ = m._train_data
        groups.setdefault((len(m.hidden_sizes), m.nonlintype, int(n_iter), int(n_batch)), []).append(m)
    dev = torch.device(device or ("cuda" if torch.cuda.is_available() else "cpu"))
    norm_dev = [torch.from_numpy(np.ascontiguousarray(v)).to(dev) for v in (xm, xs, dm, ds)]
    steps = 0
    for (depth, act, n_iter, n_batch), ms in groups.items():
        dims = [[XU.shape[1]] + list(m.hidden_sizes) + [dY.shape[1]] for m in ms]
        fit = LockstepFit(dims, act, [m._train_data[2] for m in ms], [m.seed for m in ms], feed, target, n_batch,
                          train_seeds=[train_seed] * len(ms), device=dev, use_graphs=use_graphs)
        fit.run(n_iter)
        steps = max(steps, fit.steps_done)
        for k, m in enumerate(ms):
            ws, bs = fit.parameters(k)
            m._adopt_fit(ws, bs, (xm, xs, dm, ds), norm_dev)
    if dev.type == "cuda":
        torch.cuda.synchronize(dev)
    return {"groups": len(groups), "fit_s": time.perf_counter() - t0, "steps": steps}
