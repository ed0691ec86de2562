"""The rl_games plugin seam of phc/run_hydra.py:246-268, without rl_games.

The reference hands its classes to rl_games' ``Runner`` through three factories (``algo_factory`` / ``player_factory`` /
``model_builder.network_factory``) and lets ``A2CBase.__init__`` create the vec-env from a registered creator
(``vecenv.register('RLGPU', ...)`` + ``env_configurations.register('rlgpu', ...)``, run_hydra.py:244-245).  This module provides that
surface -- same method names, same call shapes -- so the registration code of run_hydra.py works against it unchanged:

    runner = build_alg_runner()                      # run_hydra.py:248-268
    runner.load({'params': {...im.yaml...}})          # Runner.load
    runner.run({'train': True, 'play': False})        # Runner.run -> agent.train() / player.run()
"""
import copy


class ObjectFactory:
    """rl_games.common.object_factory.ObjectFactory."""

    def __init__(self):
        self._builders = {}

    def register_builder(self, name, builder):
        self._builders[name] = builder

    def set_builders(self, builders):
        self._builders = builders

    def create(self, name, **kwargs):
        builder = self._builders.get(name)
        if not builder:
            raise ValueError(name)
        return builder(**kwargs)


class _ModelBuilder:
    def __init__(self):
        self.model_factory = ObjectFactory()
        self.network_factory = ObjectFactory()


_ENV_CREATORS = {}


def register_env(name, creator):
    """env_configurations.register(name, {'env_creator': creator, ...}) + vecenv.register: ``creator(**kwargs)`` returns the
    vec-env object the agent talks to (step / reset / get_env_info)."""
    _ENV_CREATORS[name] = creator


def create_vec_env(name, num_actors, **kwargs):
    if name not in _ENV_CREATORS:
        raise ValueError(f"unknown env_name {name!r}: register it with pulse_amd.runner.register_env")
    return _ENV_CREATORS[name](num_actors=num_actors, **kwargs)


class Runner:
    """rl_games.torch_runner.Runner: load(yaml-shaped dict) then run(args)."""

    def __init__(self, algo_observer=None):
        self.algo_factory = ObjectFactory()
        self.player_factory = ObjectFactory()
        self.model_builder = _ModelBuilder()
        self.algo_observer = algo_observer
        self.params = self.config = None

    def load(self, yaml_conf):
        self.default_config = yaml_conf["params"]
        self.params = copy.deepcopy(self.default_config)
        self.algo_name = self.params["algo"]["name"]
        self.seed = self.params.get("seed", None)
        config = dict(self.params["config"])
        config["network"] = self.params["network"]          # the reference resolves params.network through model_builder.load
        config["reward_shaper"] = dict(config.get("reward_shaper", {}))
        if self.seed is not None:
            config.setdefault("seed", self.seed)
        self.config = config

    def run_train(self, max_epochs=None):
        agent = self.algo_factory.create(self.algo_name, base_name="run", config=self.config)
        self.agent = agent
        return agent.train(max_epochs=max_epochs) if max_epochs is not None else agent.train()

    def run_play(self, n_steps=None, checkpoint=None):
        player = self.player_factory.create(self.algo_name, config=self.config)
        if checkpoint is not None:
            player.restore(checkpoint)
        self.player = player
        return player.run(n_steps)

    def run(self, args):
        if args.get("train", True) and not args.get("play", False):
            return self.run_train(args.get("max_epochs"))
        return self.run_play(args.get("n_steps"), args.get("checkpoint"))


def build_alg_runner(algo_observer=None):
    """phc/run_hydra.py:246-268 with this package's classes."""
    from .learning import amp_agent, common_agent, im_amp, players
    runner = Runner(algo_observer)
    runner.algo_factory.register_builder("a2c_continuous", lambda **kw: common_agent.CommonAgent(**kw))
    runner.algo_factory.register_builder("amp", lambda **kw: amp_agent.AMPAgent(**kw))
    runner.player_factory.register_builder("amp", lambda **kw: players.AMPPlayerContinuous(**kw))
    runner.algo_factory.register_builder("im_amp", lambda **kw: im_amp.IMAmpAgent(**kw))
    runner.player_factory.register_builder("im_amp", lambda **kw: players.IMAMPPlayerContinuous(**kw))
    # the network names of the reference's network_factory: resolved inside the agents' _build_model
    for name in ("amp", "amp_z", "amp_z_reader", "amp_sept"):
        runner.model_builder.network_factory.register_builder(name, lambda **kw: None)
    return runner
