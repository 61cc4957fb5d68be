registry = {}


def register(id, entry_point=None, **kwargs):
    registry[id] = entry_point
