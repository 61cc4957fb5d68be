"""World objects of the batched engine: value-type descriptors with the reference's class names,
predicates and `encode()` (kandouss/marlgrid `marlgrid/objects.py`).

In the reference every object is a live Python instance hanging off a per-grid registry
(`base.py:19-64`).  Here an object is *data*: the env keeps one table row per distinct
`(class, colour, state, params)` and the HBM grid stores the row id (uint8) per cell; the HIP
kernels read predicates / encode triples / reward parameters from that table
(`MgObjDesc`, include/marlgrid_hip.h).  Sprites are described as fill programs
(`sprite_ops`) that `rendering.py` rasterises into the tile atlas.
"""
import numpy as np

# objects.py:11-26 — colour names -> RGB
COLORS = {
    "red": np.array([255, 0, 0]),
    "orange": np.array([255, 165, 0]),
    "green": np.array([0, 255, 0]),
    "blue": np.array([0, 0, 255]),
    "cyan": np.array([0, 139, 139]),
    "purple": np.array([112, 39, 195]),
    "yellow": np.array([255, 255, 0]),
    "olive": np.array([128, 128, 0]),
    "grey": np.array([100, 100, 100]),
    "worst": np.array([74, 65, 42]),
    "pink": np.array([255, 0, 189]),
    "white": np.array([255, 255, 255]),
    "prestige": np.array([255, 255, 255]),
    "shadow": np.array([35, 25, 30]),
}
COLOR_TO_IDX = {name: i for i, name in enumerate(COLORS)}   # objects.py:29

# encode()'s type index is the class definition order under the reference's registering
# metaclass (objects.py:31-43); GridAgentInterface (agents.py:9) lands at 13.
OBJECT_TYPES = []


class _Registered(type):
    def __new__(meta, name, bases, ns):
        cls = super().__new__(meta, name, bases, ns)
        OBJECT_TYPES.append(cls)
        return cls


def _rect(xmin, xmax, ymin, ymax, rgb):
    return ("rect", (xmin, xmax, ymin, ymax), tuple(int(v) & 255 for v in rgb))


def _circle(cx, cy, r, rgb):
    return ("circle", (cx, cy, r), tuple(int(v) & 255 for v in rgb))


class WorldObj(metaclass=_Registered):
    """objects.py:46-121.  Immutable description of one object kind."""
    overlappable = False
    pickable = False
    transparent = True
    ends_episode = False      # isinstance(fwd_cell, (Lava, Goal)) — base.py:584
    is_agent = False

    def __init__(self, color="worst", state=0):
        if color not in COLOR_TO_IDX:
            raise KeyError(color)           # the reference fails the same way at render / encode time
        self.color = color
        self.state = int(state)

    @property
    def type(self):
        return self.__class__.__name__

    @property
    def numeric_color(self):
        return COLORS[self.color]

    def can_overlap(self):
        return self.overlappable

    def can_pickup(self):
        return self.pickable

    def see_behind(self):
        return self.transparent

    def encode(self, str_class=False):
        cls = self.type if str_class else OBJECT_TYPES.index(self.__class__)
        return (cls, COLOR_TO_IDX[self.color], self.state)

    def describe(self):
        return "Obj: %s(%s, %s)" % (self.type, self.color, self.state)

    # -- engine side ---------------------------------------------------------------------------
    def key(self):
        """hashable identity of the table row this object maps to"""
        return (self.type, self.color, self.state)

    def related(self):
        """other rows that must exist whenever this one does (a Door's other states)"""
        return []

    def sprite_ops(self):
        raise NotImplementedError("%s has no sprite" % self.type)

    def __eq__(self, other):
        return isinstance(other, WorldObj) and self.key() == other.key()

    def __hash__(self):
        return hash(self.key())

    def __repr__(self):
        return "%s(color=%r, state=%r)" % (self.type, self.color, self.state)


class GridAgent(WorldObj):
    """objects.py:124-153 — registry placeholder (type index 1); agents live in `agents.py`."""
    overlappable = True
    is_agent = True

    def __init__(self, *args, color="red", **kwargs):
        super().__init__(*args, color=color, **kwargs)

    @property
    def type(self):
        return "Agent"


class BulkObj(WorldObj):
    """objects.py:156-162 (hash-by-value objects; every object is hash-by-value here)."""


class BonusTile(WorldObj):
    """objects.py:164-209"""
    overlappable = True

    def __init__(self, reward, penalty=-0.1, bonus_id=0, n_bonus=1, initial_reward=True,
                 reset_on_mistake=False, color="yellow", *args, **kwargs):
        kwargs.pop("state", None)
        super().__init__(color=color, state=bonus_id)
        self.reward, self.penalty = reward, penalty
        self.n_bonus, self.bonus_id = int(n_bonus), int(bonus_id)
        self.initial_reward, self.reset_on_mistake = bool(initial_reward), bool(reset_on_mistake)

    def key(self):
        return (self.type, self.color, self.state, float(self.reward), float(self.penalty), self.n_bonus,
                self.initial_reward, self.reset_on_mistake)

    def sprite_ops(self):
        return [_rect(0, 1, 0, 1, COLORS[self.color])]            # objects.py:208-209


class Goal(WorldObj):
    """objects.py:211-226"""
    overlappable = True
    ends_episode = True

    def __init__(self, reward, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self.reward = reward

    def key(self):
        return (self.type, self.color, self.state, float(self.reward))

    def get_reward(self, agent=None):
        return self.reward

    def sprite_ops(self):
        return [_rect(0, 1, 0, 1, COLORS[self.color])]            # objects.py:225-226


class Floor(WorldObj):
    """objects.py:229-246 (its render uses a removed API upstream: no sprite)."""
    overlappable = True


class EmptySpace(WorldObj):
    """objects.py:249-254 (`can_verlap` typo upstream: not overlappable; no render)."""


class Lava(WorldObj):
    """objects.py:257-277 (render raises NameError upstream: no sprite)."""
    overlappable = True
    ends_episode = True


class Wall(BulkObj):
    """objects.py:280-288"""
    transparent = False

    def sprite_ops(self):
        return [_rect(0, 1, 0, 1, COLORS[self.color])]


class Key(WorldObj):
    """objects.py:291-310.  Upstream's render raises NameError (`point_in_circle` is never
    imported); the sprite below is the drawing that code spells out."""
    pickable = True

    def sprite_ops(self):
        c = COLORS[self.color]
        return [_rect(0.50, 0.63, 0.31, 0.88, c), _rect(0.38, 0.50, 0.59, 0.66, c),
                _rect(0.38, 0.50, 0.81, 0.88, c), _circle(0.56, 0.28, 0.190, c),
                _circle(0.56, 0.28, 0.064, (0, 0, 0))]


class Ball(WorldObj):
    """objects.py:313-321 (same NameError upstream)."""
    pickable = True

    def sprite_ops(self):
        return [_circle(0.5, 0.5, 0.31, COLORS[self.color])]


class Door(WorldObj):
    """objects.py:324-370; states open=1, closed=2, locked=3 (objects.py:325)."""
    OPEN, CLOSED, LOCKED = 1, 2, 3

    class states(object):
        open, closed, locked = 1, 2, 3

    def __init__(self, color="worst", state=2):
        super().__init__(color=color, state=int(state))
        if self.state not in (1, 2, 3):
            raise ValueError("Door state must be 1 (open), 2 (closed) or 3 (locked)")

    def can_overlap(self):
        return self.state == self.OPEN

    def see_behind(self):
        return self.state == self.OPEN

    def related(self):
        return [Door(self.color, s) for s in (1, 2, 3) if s != self.state]

    def sprite_ops(self):
        c = COLORS[self.color]
        if self.state == self.OPEN:
            return [_rect(0.88, 1.00, 0.00, 1.00, c), _rect(0.92, 0.96, 0.04, 0.96, (0, 0, 0))]
        if self.state == self.LOCKED:
            dim = 0.45 * np.array(c)                # float colour, truncated on uint8 assignment
            return [_rect(0.00, 1.00, 0.00, 1.00, c), _rect(0.06, 0.94, 0.06, 0.94, dim),
                    _rect(0.52, 0.75, 0.50, 0.56, c)]
        # closed (NameError upstream, objects.py:370): the drawing that code spells out
        return [_rect(0.00, 1.00, 0.00, 1.00, c), _rect(0.04, 0.96, 0.04, 0.96, (0, 0, 0)),
                _rect(0.08, 0.92, 0.08, 0.92, c), _rect(0.12, 0.88, 0.12, 0.88, (0, 0, 0)),
                _circle(0.75, 0.50, 0.08, c)]


class Box(WorldObj):
    """objects.py:373-395 (toggle() has the wrong arity upstream -> TypeError when toggled)."""
    pickable = True

    def __init__(self, color=0, state=0, contains=None):
        super().__init__(color, state)      # default colour 0 raises KeyError, as upstream does later
        self.contains = contains

    def sprite_ops(self):
        c = COLORS[self.color]
        return [_rect(0.12, 0.88, 0.12, 0.88, c), _rect(0.18, 0.82, 0.18, 0.82, (0, 0, 0)),
                _rect(0.16, 0.84, 0.47, 0.53, c)]
