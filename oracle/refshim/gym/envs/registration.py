class EnvSpec:
    def __init__(self, id, max_episode_steps=None, **kw):
        self.id = id
        self.max_episode_steps = max_episode_steps
