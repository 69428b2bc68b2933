"""Karel grid world: state tensor [h, w, 16] of booleans, five actions, five perceptions.

Restates the behaviour of the reference world (karel_env/karel.py:34-185) that the execution
metric relies on (models/model_full.py:745-780): channel layout 0-3 heading N/E/S/W, 4 wall,
5+n = n markers; `move` into a wall or the border raises when make_error, otherwise turns Karel
around; pick below 0 / put above 9 markers raise when make_error, otherwise leave the count;
every executed action appends the new state to `s_h` and its index to `a_h`.
"""
import numpy as np

MAX_NUM_MARKER = 10
ACTIONS = ('move', 'turnLeft', 'turnRight', 'pickMarker', 'putMarker')
PERCEPTIONS = ('frontIsClear', 'leftIsClear', 'rightIsClear', 'markersPresent', 'noMarkersPresent')

# heading -> (dy, dx) of the cell in front; left / right are the neighbouring headings
_FRONT = ((-1, 0), (0, 1), (1, 0), (0, -1))


class Karel_world(object):
    """Same constructor and attribute names as the reference class (`s`, `s_h`, `a_h`, `p_v_h`)."""

    def __init__(self, s=None, make_error=True):
        self.make_error = make_error
        if s is not None:
            self.set_new_state(s)

    def set_new_state(self, s):
        self.s = np.array(s).astype(bool)
        self.h, self.w = self.s.shape[:2]
        self.s_h = [self.s.copy()]
        self.a_h = []
        self.p_v_h = [self.get_perception_vector()]

    def clear_history(self):
        self.s_h = [self.s.copy()]
        self.a_h = []

    # ------------------------------------------------------------------ geometry
    def get_location(self):
        """(row, col, heading) of Karel: first set bit of channels 0..3 in row-major order."""
        idx = np.flatnonzero(self.s[:, :, :4])
        if idx.size == 0:
            raise IndexError('no Karel in the world')
        y, rem = divmod(int(idx[0]), self.w * 4)
        x, z = divmod(rem, 4)
        return y, x, z

    def _neighbor(self, turn):
        """Cell one step towards heading + turn (0 front, -1 left, +1 right)."""
        y, x, z = self.get_location()
        dy, dx = _FRONT[(z + turn) % 4]
        return y + dy, x + dx

    def _clear(self, turn):
        ny, nx = self._neighbor(turn)
        if ny < 0 or ny >= self.h or nx < 0 or nx >= self.w:
            return False
        return not self.s[ny, nx, 4]

    def front_is_clear(self):
        return self._clear(0)

    def left_is_clear(self):
        return self._clear(-1)

    def right_is_clear(self):
        return self._clear(1)

    def marker_present(self):
        y, x, _ = self.get_location()
        return bool(self.s[y, x, 6:].any())

    def no_marker_present(self):
        return not self.marker_present()

    def get_perception_list(self):
        return list(PERCEPTIONS)

    def get_perception_vector(self):
        return np.array([self.front_is_clear(), self.left_is_clear(), self.right_is_clear(),
                         self.marker_present(), self.no_marker_present()])

    # ------------------------------------------------------------------ transitions
    def _record(self, a_idx):
        self.s_h.append(self.s.copy())
        self.a_h.append(a_idx)
        self.p_v_h.append(self.get_perception_vector())

    def state_transition(self, a):
        """`a`: one-hot (or score) vector over the five actions, or an int index."""
        a_idx = int(a) if np.isscalar(a) else int(np.argmax(a))
        y, x, z = self.get_location()
        if a_idx == 0:
            if self.front_is_clear():
                ny, nx = self._neighbor(0)
                self.s[ny, nx, :4] = self.s[y, x, :4]
                self.s[y, x, :4] = False
            else:
                if self.make_error:
                    raise RuntimeError('Failed to move.')
                self.s[y, x, :4] = False
                self.s[y, x, (z + 2) % 4] = True
        elif a_idx in (1, 2):
            self.s[y, x, :4] = False
            self.s[y, x, (z + (-1 if a_idx == 1 else 1)) % 4] = True
        elif a_idx in (3, 4):
            n = int(np.argmax(self.s[y, x, 5:]))
            new_n = n + (-1 if a_idx == 3 else 1)
            if new_n < 0:
                if self.make_error:
                    raise RuntimeError('No marker to pick up.')
                new_n = n
            elif new_n > MAX_NUM_MARKER - 1:
                if self.make_error:
                    raise RuntimeError('Cannot put more marker.')
                new_n = n
            self.s[y, x, 5:] = False
            self.s[y, x, 5 + new_n] = True
        else:
            raise RuntimeError('Invalid action')
        self._record(a_idx)
