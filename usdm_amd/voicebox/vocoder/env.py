"""AttrDict config bag, as vocoder/env.py:8-11 of the reference."""


class AttrDict(dict):
    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self.__dict__ = self
