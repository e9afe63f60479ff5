"""On-device rollout: the batched counterpart of MAPPO.interact / evaluation.

The reference collects experience one env-step at a time on the host (marl/mappo.py:102-158): per
agent a shared actor MLP forward, `np.random.choice` over the softmax, `env.step`, regional rewards
scaled by `reward_scale`, and a discounted return bootstrapped from the critic when the rollout
ends mid-episode (`_discount_reward`, :364-370).  Here the same quantities are produced for E
envs at once with every tensor staying in HBM: observations come straight out of `mm_step`, the
actor forward + categorical sample is ONE launch (`mm_policy_act`: f32-input MFMA, inverse-CDF sampling exactly
as `np.random.choice` does it, Philox uniforms; other actor modules run their own forward followed by
`mm_sample_actions`), and the discounting is a reversed scan over the rollout with episode boundaries taken from
`done`.

The loop is launch-bound once `mm_step` takes < 0.5 ms (about a dozen small launches per policy
step), so `use_graph=True` captures the WHOLE rollout -- T x (policy forward, sample, mm_step,
bookkeeping) + bootstrap + discounting -- into one hipGraph on first use and replays it afterwards
(static output buffers; the env state, the carried observation and the RNG offset advance inside
the graph exactly as they do eagerly).

Networks mirror marl/single_agent/Model_common.py:5-41 (state -> 128 -> 128 -> n_a, log-softmax;
critic takes the one-hot action after the first layer).
"""
import torch
from torch import nn

from . import _cabi as abi


class ActorNetwork(nn.Module):
    """Model_common.py:5-22."""

    def __init__(self, state_dim, hidden_size, output_size):
        super().__init__()
        self.fc1 = nn.Linear(state_dim, hidden_size)
        self.fc2 = nn.Linear(hidden_size, hidden_size)
        self.fc3 = nn.Linear(hidden_size, output_size)

    def forward(self, state):
        out = torch.relu(self.fc1(state))
        out = torch.relu(self.fc2(out))
        return torch.log_softmax(self.fc3(out), dim=-1)


class CriticNetwork(nn.Module):
    """Model_common.py:25-41."""

    def __init__(self, state_dim, action_dim, hidden_size, output_size=1):
        super().__init__()
        self.fc1 = nn.Linear(state_dim, hidden_size)
        self.fc2 = nn.Linear(hidden_size + action_dim, hidden_size)
        self.fc3 = nn.Linear(hidden_size, output_size)

    def forward(self, state, action_one_hot):
        out = torch.relu(self.fc1(state))
        out = torch.relu(self.fc2(torch.cat([out, action_one_hot], dim=-1)))
        return self.fc3(out)


def discount_rewards(rewards, dones, final_value, gamma):
    """`_discount_reward` (marl/mappo.py:364-370) over a batch.

    rewards [T, E, N], dones [T, E] (episode ended AT step t), final_value [E, N] (bootstrap for the
    envs whose last step did not end an episode; the reference uses 0 after `done`).
    running = final; for t reversed: running = running * gamma + r[t]; a `done` at step t cuts the
    chain coming from later steps (they belong to the next episode of that env slot)."""
    T = rewards.shape[0]
    out = torch.empty_like(rewards)
    running = final_value.clone()
    for t in range(T - 1, -1, -1):
        running = torch.where(dones[t].bool().unsqueeze(-1), torch.zeros_like(running), running)
        running = running * gamma + rewards[t]
        out[t] = running
    return out


class DeviceRollout(object):
    """MAPPO.interact for a whole env batch; see the module docstring."""

    def __init__(self, env, actor, critic=None, roll_out_n_steps=100, reward_gamma=0.99, reward_scale=20.0,
                 reward_type="regionalR", generator=None, use_graph=False, sample_seed=0, fused_policy=True):
        assert reward_type in ("regionalR", "global_R")  # marl/mappo.py:39
        self.env, self.actor, self.critic = env, actor, critic
        self.T, self.gamma, self.reward_scale, self.reward_type = roll_out_n_steps, reward_gamma, reward_scale, reward_type
        self.generator = generator
        self.n_a = env.n_a
        self.sample_seed = int(sample_seed) & 0xFFFFFFFFFFFFFFFF
        # fused actor + sampling launch for the reference's ActorNetwork (hidden 128); anything else goes
        # through the module's own forward
        self.fused_policy = bool(fused_policy) and type(actor) is ActorNetwork and actor.fc2.weight.shape[0] == 128 \
            and next(actor.parameters()).dtype == torch.float32
        self.obs, _ = env.reset()
        self.obs = self.obs.clone()
        self._sample_counter = torch.zeros(1, dtype=torch.int64, device=self.obs.device)  # advanced by mm_sample_actions
        self.use_graph = bool(use_graph)
        if self.use_graph:
            if self.obs.device.type != "cuda":
                raise RuntimeError("use_graph needs the device backend (hipGraph capture)")
            if generator is not None:
                raise ValueError("use_graph samples from the default CUDA generator (captured Philox offset)")
        self._graph, self._static = None, None

    def state_dict(self):
        """Resume point of a rollout stream: env batch + carried observation + sampler counter."""
        return {"env": self.env.state_dict(), "obs": self.obs.clone(), "sample_counter": self._sample_counter.clone(),
                "sample_seed": self.sample_seed}

    def load_state_dict(self, d):
        self.env.load_state_dict(d["env"])
        self.obs.copy_(d["obs"].to(self.obs.device))  # in place: with a captured graph self.obs IS its static carry buffer
        self._sample_counter.copy_(d["sample_counter"].to(self._sample_counter.device))
        if int(d["sample_seed"]) != self.sample_seed:
            self.sample_seed = int(d["sample_seed"])
            self._graph, self._static = None, None  # the seed is a kernel argument baked into the capture: re-capture

    @torch.no_grad()
    def act(self, obs, out=None):
        """exploration_action / action (marl/mappo.py:220-236): sample from softmax(actor(obs)).
        out: optional int32 [E, N] buffer (contiguous, on the device) the actions are written to -- a rollout's actions[t]."""
        if out is not None:
            assert out.shape == obs.shape[:2] and out.dtype == torch.int32 and out.device == obs.device and out.is_contiguous()
            return self._act(obs, out.view(-1))
        return self._act(obs, None)

    def _act(self, obs, dst):
        E, N, S = obs.shape
        if self.generator is None and self.fused_policy and type(self.actor) is ActorNetwork and obs.dtype == torch.float32:
            # actor forward + sampling in ONE launch (mm_policy_act: f32 MFMA, activations in registers)
            a, clib = self.actor, self.env.clib
            actions = dst if dst is not None else torch.empty(E * N, dtype=torch.int32, device=obs.device)
            ptr = lambda t: t.detach().contiguous().data_ptr()  # noqa: E731  (nn.Linear parameters are contiguous)
            clib.check(clib.lib.mm_policy_act(obs.contiguous().data_ptr(), E * N, S, ptr(a.fc1.weight), ptr(a.fc1.bias),
                                              ptr(a.fc2.weight), ptr(a.fc2.bias), ptr(a.fc3.weight), ptr(a.fc3.bias),
                                              a.fc2.weight.shape[0], self.n_a, self.sample_seed,
                                              self._sample_counter.data_ptr(), actions.data_ptr(), None, self.env._stream()))
            return actions.view(E, N)
        logp = self.actor(obs.reshape(E * N, S).float())
        # np.random.choice(n_a, p=softmax) (marl/mappo.py:229) is inverse-CDF sampling: cdf.searchsorted(u, "right").
        if self.generator is None:
            # fused in the library (mm_sample_actions): softmax -> cdf -> search, Philox uniform per agent, one
            # pass over 24 B/agent instead of five elementwise / scan kernels over fp64 temporaries
            logp = logp.contiguous()
            actions = dst if dst is not None else torch.empty(E * N, dtype=torch.int32, device=obs.device)
            clib = self.env.clib
            stream = self.env._stream()
            clib.check(clib.lib.mm_sample_actions(logp.data_ptr(), E * N, self.n_a, self.sample_seed,
                                                  self._sample_counter.data_ptr(), actions.data_ptr(), stream))
            return actions.view(E, N)
        # caller-supplied torch generator: same arithmetic with torch ops
        cdf = logp.exp().double().cumsum(-1)
        cdf = cdf / cdf[:, -1:]
        u = torch.rand(E * N, 1, dtype=torch.float64, device=obs.device, generator=self.generator)
        a = (cdf <= u).sum(-1).clamp_(max=self.n_a - 1).to(torch.int32)
        if dst is not None:
            dst.copy_(a)
            a = dst
        return a.view(E, N)

    @torch.no_grad()
    def interact(self):
        """One rollout of `roll_out_n_steps` policy steps on every env (auto-reset on).
        Returns states [T,E,N,S], actions [T,E,N], discounted returns [T,E,N], dones [T,E] and the
        per-step info means the reference logs (average speed, min headway).  With `use_graph` the
        returned tensors are the graph's static buffers: consume them before the next call."""
        if not self.use_graph:
            return self._interact()
        if self._graph is None:
            dev = self.obs.device
            if not getattr(self, "_warmed", False):
                # first capture only: one eager rollout outside capture (lazy library init, allocator warm-up).  It is a
                # real rollout of the stream (state, RNG counter advance); its tensors are simply not returned.
                self._carry = self.obs  # static input/output of the graph: the observation carried between rollouts
                side = torch.cuda.Stream(device=dev)
                side.wait_stream(torch.cuda.current_stream(dev))
                with torch.cuda.stream(side):
                    self._interact()
                    self._carry.copy_(self.obs)
                    self.obs = self._carry
                torch.cuda.current_stream(dev).wait_stream(side)
                self._warmed = True
            torch.cuda.synchronize(dev)
            self._graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self._graph):
                self._static = self._interact()
                self._carry.copy_(self.obs)
            self.obs = self._carry
        self._graph.replay()
        return self._static

    @torch.no_grad()
    def _interact(self):
        env, T = self.env, self.T
        E, N, S = self.obs.shape
        dev = self.obs.device
        states = torch.empty(T + 1, E, N, S, dtype=self.obs.dtype, device=dev)  # slot t + 1 is written by step t itself
        actions = torch.empty(T, E, N, dtype=torch.int32, device=dev)
        rewards = torch.empty(T, E, N, dtype=torch.float64, device=dev)
        dones = torch.empty(T, E, dtype=torch.uint8, device=dev)
        speeds = torch.empty(T, E, dtype=torch.float64, device=dev)
        headways = torch.empty(T, E, dtype=torch.float64, device=dev)
        states[0] = self.obs
        obs = states[0]
        regional = self.reward_type == "regionalR"
        for t in range(T):
            # every per-step quantity lands in its [t] slot straight from the two launches (policy + step): no copy kernels
            a = self.act(obs, out=actions[t])
            slots = {"done": dones[t], "average_speed": speeds[t], "min_headway": headways[t]}
            if regional:
                slots["regional_rewards"] = rewards[t]
            obs, global_reward, _, _ = env.step(a, obs_out=states[t + 1], out=slots)
            if not regional:
                rewards[t] = global_reward.unsqueeze(-1).expand(E, N)
        speed_sum = speeds.sum(0)
        min_headway = headways.min(0).values
        self.obs = obs.clone()
        # bootstrap value for envs still mid-episode (marl/mappo.py:147-150); 0 where the last step ended one
        final_value = torch.zeros(E, N, dtype=torch.float64, device=dev)
        if self.critic is not None:
            fa = self.act(obs)
            # (not F.one_hot: its device-side range asserts do not survive hipGraph capture)
            one_hot = (fa.unsqueeze(-1) == torch.arange(self.n_a, device=dev, dtype=fa.dtype)).float()
            val = self.critic(obs.reshape(E * N, S).float(), one_hot.view(E * N, self.n_a)).view(E, N).double()
            final_value = torch.where(dones[-1].bool().unsqueeze(-1), final_value, val)
        # reward scaling (:152-153) + _discount_reward (:364-370) in one launch (mm_discount_returns; `discount_rewards`
        # above is the same arithmetic in torch ops, 3 launches per step of the rollout)
        returns = torch.empty_like(rewards)
        clib = env.clib  # (errors the reference raises inside step() are latched on the device: the training loop polls them,
        #                   env.poll_errors(), where it synchronises anyway -- a poll per rollout here would break hipGraph capture)
        clib.check(clib.lib.mm_discount_returns(rewards.data_ptr(), dones.data_ptr(), final_value.contiguous().data_ptr(), T, E, N,
                                                float(self.gamma), float(self.reward_scale), returns.data_ptr(), env._stream()))
        return {"states": states[:T], "actions": actions, "returns": returns, "dones": dones,
                "average_speed": speed_sum / T, "min_headway": min_headway}

    @torch.no_grad()
    def evaluate(self, seeds=None):
        """MAPPO.evaluation (marl/mappo.py:255-361) for a whole batch: every env slot runs ONE evaluation
        episode (its own seed, `env.reset(is_training=False, testing_seeds=seed)` in the reference) to its
        terminal step while the others keep stepping; per-episode results are taken at each env's own end.

        Returns (rewards [T_max, E] with NaN after an episode's end, (vehicle_speed, vehicle_position)
        [T_max, E, N] likewise, ext_info) where ext_info has the reference's keys: steps, avg_speeds,
        crash_count, min_headway (one number, min over all steps of all episodes), traffic_speeds,
        merge_percents -- per env as tensors."""
        env = self.env
        E, N, dev = env.E, env.N, self.obs.device
        missing = [k for k in ("agents_info", "crashed") if k not in env.out]
        if missing:  # MAPPO.evaluation reads info["vehicle_speed"/"vehicle_position"] and env.is_crashed() (:300-330)
            raise ValueError("evaluate() needs the step outputs %s: this env was built with skip_outputs" % ", ".join(missing))
        was_auto = env.auto_reset
        # The reference evaluates on a SEPARATE env (env_eval, run_mappo.py:146-171,300-306) and its training env keeps
        # its own seed sequence and in-progress episode.  Here the batch is borrowed: snapshot everything an evaluation
        # touches (state planes, episode counters, per-env RNG seeds, carried observation) and put it back afterwards,
        # so the training stream continues exactly where it was instead of restarting from the test seeds.
        saved_env, saved_obs, saved_counter = env.state_dict(), self.obs.clone(), self._sample_counter.clone()
        env.configure(auto_reset=False)
        try:
            # an evaluation episode is a function of its seed alone (the reference re-seeds the global RNG):
            # restart the per-slot episode counter that the device RNG mixes in
            env.env_i32[abi.EP["EPISODE"]].zero_()
            obs, _ = env.reset(seeds=None if seeds is None else torch.as_tensor(seeds, dtype=torch.int64))
            T = env.T
            nan = float("nan")
            rewards = torch.full((T, E), nan, dtype=torch.float64, device=dev)
            vspeed = torch.full((T, E, N), nan, dtype=torch.float64, device=dev)
            vpos = torch.full((T, E, N), nan, dtype=torch.float64, device=dev)
            alive = torch.ones(E, dtype=torch.bool, device=dev)
            steps = torch.zeros(E, dtype=torch.int32, device=dev)
            speed_sum = torch.zeros(E, dtype=torch.float64, device=dev)
            tspeed_sum = torch.zeros(E, dtype=torch.float64, device=dev)
            crash = torch.zeros(E, dtype=torch.bool, device=dev)
            merge = torch.full((E,), nan, dtype=torch.float64, device=dev)
            min_headway = torch.full((), float("inf"), dtype=torch.float64, device=dev)
            for t in range(T):
                a = self.act(obs)
                obs, reward, done, info = env.step(a)
                rewards[t] = torch.where(alive, reward, rewards[t])
                vspeed[t] = torch.where(alive[:, None], info["agents_info"][..., 2], vspeed[t])
                vpos[t] = torch.where(alive[:, None], info["agents_info"][..., 0], vpos[t])
                steps += alive.int()
                speed_sum += torch.where(alive, info["average_speed"], torch.zeros_like(speed_sum))
                tspeed_sum += torch.where(alive, info["traffic_speed"], torch.zeros_like(tspeed_sum))
                mh = torch.where(alive, info["min_headway"], torch.full_like(info["min_headway"], float("inf")))
                min_headway = torch.minimum(min_headway, mh.min())
                ending = alive & done.bool()
                crash |= ending & info["crashed"].bool().any(-1)  # env.is_crashed() at the episode's end
                merge = torch.where(ending, info["merge_percent"], merge)
                alive = alive & ~done.bool()
                if not bool(alive.any()):  # one host sync per policy step; evaluation is not the hot loop
                    break
            n = steps.clamp(min=1).double()
            ext_info = {"steps": steps, "avg_speeds": speed_sum / n, "crash_count": crash, "min_headway": float(min_headway),
                        "traffic_speeds": tspeed_sum / n, "merge_percents": merge}
            tmax = int(steps.max())
            return rewards[:tmax], (vspeed[:tmax], vpos[:tmax]), ext_info
        finally:
            env.configure(auto_reset=was_auto)
            env.load_state_dict(saved_env)
            # in place: after a graph capture self.obs is the graph's static carry buffer and must stay that tensor
            self.obs.copy_(saved_obs)
            # the sampler's counter as well (the evaluation's own act() calls advanced it): with a stochastic actor the
            # training rollouts after an evaluation are then the ones a twin that never evaluated draws
            self._sample_counter.copy_(saved_counter)
