"""AttrDict — the config tree's node type: a dict whose keys read and write as attributes (`cfg.TRAIN.MAX_SIZE`), the
behaviour of reference lib/utils/collections.py:15-27.  Plain dict storage, so yaml / pickle / deepcopy see a dict."""


class AttrDict(dict):
    __slots__ = ()

    def __getattr__(self, key):
        try:
            return self[key]
        except KeyError:
            raise AttributeError(key)

    def __setattr__(self, key, value):
        self[key] = value

    def __delattr__(self, key):
        try:
            del self[key]
        except KeyError:
            raise AttributeError(key)
