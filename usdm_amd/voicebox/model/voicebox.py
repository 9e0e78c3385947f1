"""CFM solvers + Voicebox drop-in (reference: src/decoder/voicebox/model/voicebox.py:18-176).

Same class names, constructor, `.generate(...)` signature, `.estimator`, `.n_tokens`, state-dict keys
(`estimator.*`).  Additions: `noise=` lets the caller supply the N(0,1) draws the reference takes
with randn_like (device RNG != CPU RNG; the parity tests feed the reference's draws), and
`from_pretrained` loads from a local directory (config.json + model.safetensors / pytorch_model.bin).
The estimator evaluation is one hipGraph replay; the solver update is one fused elementwise kernel.
"""
import json
import os

import torch
from torch import nn

from ... import ops
from .networks import Transformer


class BaseModule(nn.Module):
    @property
    def nparams(self):
        return sum(p.numel() for p in self.parameters() if p.requires_grad)


class CFM(BaseModule):
    def __init__(self, solver, sigma_min):
        super().__init__()
        self.solver = solver
        self.sigma_min = sigma_min
        self.estimator = None

    def forward(self, x, mask, x1, lengths):
        raise NotImplementedError("training loss (voicebox.py:28-49) is outside the inference hot path")

    @torch.no_grad()
    def generate(self, x, cond, cond_lengths, n_timesteps, solver="euler", gradient_scale=0.0, speech_prompt=False,
                 prompt_lengths=None, noise=None, trace=None, cfg_group=None):
        """x int64 [B,S], cond f32 [B,80,S], cond_lengths int64 [B] -> f32 [B,80,S]
        (reference: voicebox.py:140-150 with solve_euler :74-99 / solve_heun :101-138 and the CFG of :51-72).
        trace (tests only): a list that receives a copy of the raw estimator output of every NFE.
        cfg_group (optional): a torch.distributed group of exactly TWO ranks that hold the same model and are called
        with the same arguments (and the same `noise`).  The classifier-free-guidance doubling of voicebox.py:60-65 is then split
        over the pair: group rank 0 evaluates the unconditional half (null tokens, zero cond), group rank 1 the conditional half,
        each at batch B instead of 2B; one all-gather of the [B,80,S] velocities per NFE rebuilds the [2B,80,S] estimator output
        and both ranks apply the same solver update (v_c + gs (v_c - v_u)), so both return the same mel."""
        if solver not in ("euler", "heun"):
            return None  # the reference falls through and returns None for unknown solvers
        if not cond.is_cuda:
            raise RuntimeError("Voicebox.generate (usdm_amd) runs on the MI355X only; there is no CPU fallback")
        dev = cond.device
        B, F_, S = cond.shape
        lens = cond_lengths.to("cpu")
        if bool((lens > S).any()) or bool((lens < 0).any()):
            raise ValueError("cond_lengths must lie in [0, frames]")
        ragged = not bool((lens == S).all())
        heun = solver == "heun"
        n = (n_timesteps + 1) // 2 if heun else n_timesteps
        P = int(prompt_lengths[0]) if speech_prompt else 0
        n_noise = 1 + ((2 * n - 1 if heun else n) if speech_prompt else 0)
        if noise is None:
            noise = torch.randn(n_noise, B, F_, S, device=dev, dtype=torch.float32)
        else:
            noise = (torch.stack(list(noise)) if not torch.is_tensor(noise) else noise).to(dev, torch.float32).contiguous()
            if noise.shape != (n_noise, B, F_, S):
                raise ValueError(f"noise must have shape {(n_noise, B, F_, S)}, got {tuple(noise.shape)}")
        noise_in = noise
        cfg = gradient_scale > 0
        # One plan / hipGraph / workspace per BUCKET of lengths (Transformer.bucket_frames): the state, the noise and the
        # condition are zero-padded to the bucket's frame count Sb and the true lengths go to the kernels through kv_len.
        Sb = self.estimator.bucket_frames(S)
        ragged = ragged or Sb != S
        split = cfg and cfg_group is not None
        half, gather = 0, None
        if split:
            import torch.distributed as dist
            if dist.get_world_size(cfg_group) != 2:
                raise ValueError("cfg_group must hold exactly two ranks (unconditional half, conditional half)")
            half = dist.get_rank(cfg_group)
            staged = dist.get_backend(cfg_group) == "gloo"        # validation runs (ranks sharing a GPU): through host memory

            def gather(dst, src):
                if staged:
                    c = torch.empty(dst.shape, dtype=dst.dtype)
                    dist.all_gather_into_tensor(c, src.cpu(), group=cfg_group)
                    dst.copy_(c)
                else:
                    dist.all_gather_into_tensor(dst, src, group=cfg_group)
        # the unconditional half is the estimator on null tokens with a zero condition (voicebox.py:60-65): the plan of a split
        # rank is the batch-B plan with (use_cond = False, ids = null) or (use_cond as given, real ids)
        use_cond = bool(speech_prompt) and not (split and half == 0)
        gp, io = self.estimator.get_plan(B, Sb, 2 if (cfg and not split) else 1, use_cond, dev, ragged)
        vl = (cond_lengths + 1).to(torch.int32)
        io["kv_len"].copy_(torch.cat([vl, vl]) if (cfg and not split) else vl)
        if Sb != S:
            io["ids"].zero_(); io["cond"].zero_()
            noise = torch.nn.functional.pad(noise, (0, Sb - S))
        if split and half == 0:
            io["ids"].fill_(self.n_tokens)                    # the null token (estimator vocabulary = n_tokens + 1)
        else:
            io["ids"][:, :S].copy_(x)
        # the solver's cond (prompt re-noising, voicebox.py:115-117) is the caller's cond on BOTH ranks; a split rank keeps it
        # apart from the estimator's cond input, which is never read on the unconditional rank (use_cond = False)
        if split:
            condf = torch.zeros_like(io["cond"])
            condf[:, :, :S].copy_(cond)
            io["cond"].copy_(condf)
            vout2 = torch.zeros(2 * B, F_, Sb, device=dev, dtype=torch.float32)
        else:
            io["cond"][:, :, :S].copy_(cond)
            condf = io["cond"]
        Z = noise[0].clone()
        v1 = torch.empty_like(Z)
        io["y"].copy_(Z)
        S_true, S = S, Sb
        # time grid on the host in fp32, computed exactly as the reference does (voicebox.py:145,102,135)
        t_span = torch.linspace(0, 1, n + 1)
        t, dt = t_span[0], t_span[1] - t_span[0]
        io["t"].fill_(float(t))
        Bx = B * (2 if (cfg and not split) else 1)       # batch rows of THIS rank's estimator plan (their time inputs)
        k = 1
        common = dict(B=B, F=F_, S=S, cfg=cfg, gs=float(gradient_scale), z_in=io["y"], t_cur=io["t"], t_count=Bx, cond=condf, P=P)
        def estimator_out():
            gp.run()
            if not split:
                return io["out"]
            gather(vout2, io["out"])                          # [uncond ; cond], the order the solver kernel's CFG combine expects
            return vout2

        for steps in range(1, n + 1):
            vout = estimator_out()
            if trace is not None:
                trace.append(vout[:, :, :S_true].clone())
            t = t + dt
            c_eps, c_cond = float(1 - (1 - self.sigma_min) * t), float(t)
            last = steps == n
            do_corr = heun and not last
            eps = None
            if speech_prompt:
                eps, k = noise[k], k + 1
            ops.vb_solver_step(vout, Z, mode=0, dt=float(dt), v1=v1, eps=eps, c_eps=c_eps, c_cond=c_cond,
                               z_commit=None if do_corr else Z, t_next=float(t), **common)
            if do_corr:
                vout = estimator_out()
                if trace is not None:
                    trace.append(vout[:, :, :S_true].clone())
                eps = None
                if speech_prompt:
                    eps, k = noise[k], k + 1
                ops.vb_solver_step(vout, Z, mode=1, dt=float(dt), v1=v1, eps=eps, c_eps=c_eps, c_cond=c_cond,
                                   z_commit=Z, t_next=float(t), **common)
            if not last:
                dt = t_span[steps + 1] - t
        # Folded LayerNorm guard (networks.LN_GUARD_RATIO): an input that left the fold's accuracy bound switches the fold off and the
        # whole solve is repeated on the LayerNorm-kernel plan with the same draws.  A CFG pair decides together.
        tripped = self.estimator.ln_guard_tripped()
        if split:
            flags = torch.zeros(2, device=dev, dtype=torch.float32)
            gather(flags, torch.full((1,), float(tripped), device=dev))
            tripped = bool(flags.sum().item() > 0)
            if tripped:                      # the partner's half tripped it: same decision on both ranks
                self.estimator.ln_fold_ok = False
                self.estimator._plans.clear()
        if tripped:
            if trace is not None:
                del trace[:]
            return self.generate(x, cond, cond_lengths, n_timesteps, solver=solver, gradient_scale=gradient_scale, speech_prompt=speech_prompt,
                                 prompt_lengths=prompt_lengths, noise=noise_in, trace=trace, cfg_group=cfg_group)
        return Z[:, :, :S_true].contiguous() if S_true != S else Z


class Voicebox(CFM):
    def __init__(self, n_feats, n_tokens, embedding_dim, hidden_size, intermediate_size, num_attention_heads,
                 num_hidden_layers, convpos_width, convpos_groups, convpos_depth, attention_dropout, activation_dropout,
                 hidden_dropout, solver, sigma_min):
        super().__init__(solver=solver, sigma_min=sigma_min)
        self.n_tokens = n_tokens
        self._init_kwargs = dict(n_feats=n_feats, n_tokens=n_tokens, embedding_dim=embedding_dim, hidden_size=hidden_size,
                                 intermediate_size=intermediate_size, num_attention_heads=num_attention_heads,
                                 num_hidden_layers=num_hidden_layers, convpos_width=convpos_width, convpos_groups=convpos_groups,
                                 convpos_depth=convpos_depth, attention_dropout=attention_dropout,
                                 activation_dropout=activation_dropout, hidden_dropout=hidden_dropout, solver=solver,
                                 sigma_min=sigma_min)
        self.estimator = Transformer(n_feats, n_tokens + 1, embedding_dim, hidden_size, intermediate_size, num_attention_heads,
                                     num_hidden_layers, convpos_width, convpos_groups, convpos_depth, attention_dropout,
                                     activation_dropout, hidden_dropout)

    # hub-mixin surface, local directories only (no network in this environment)
    @classmethod
    def from_pretrained(cls, pretrained_model_name_or_path, cache_dir=None, **kw):
        d = pretrained_model_name_or_path
        if not os.path.isdir(d) and cache_dir is not None:      # the reference's call: hub name + cache_dir (model_util.py:59-62)
            from ...checkpoints import resolve_local
            d = resolve_local(cache_dir, d, ("config.json",))
        if not os.path.isdir(d):
            raise FileNotFoundError(f"{d}: only local directories can be loaded (no network); expected config.json + weights")
        with open(os.path.join(d, "config.json")) as f:
            cfg = json.load(f)
        model = cls(**cfg)
        st = os.path.join(d, "model.safetensors")
        if os.path.exists(st):
            from safetensors.torch import load_file
            sd = load_file(st)
        else:
            sd = torch.load(os.path.join(d, "pytorch_model.bin"), map_location="cpu")
        model.load_state_dict(sd)
        return model

    def save_pretrained(self, save_directory):
        os.makedirs(save_directory, exist_ok=True)
        with open(os.path.join(save_directory, "config.json"), "w") as f:
            json.dump(self._init_kwargs, f, indent=2)
        from safetensors.torch import save_file
        save_file({k: v.contiguous() for k, v in self.state_dict().items()}, os.path.join(save_directory, "model.safetensors"))
