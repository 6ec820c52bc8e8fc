"""Edits of the canopy STATE the reference's SAILH reads from the object at call time -- ``canopy.lidf`` (sailh.py:51) and
``canopy.nlayers`` (sailh.py:48) -- shared by the fixture generator (make_golden.py gen_canopy_state: applied to the
reference's CanopyStructure) and by the tests (applied to this package's).  Pure numpy on attributes: nothing of either
implementation is imported here.

The reference's constructor evaluates ``lidf = calculate_leafangles(LIDFa, LIDFb)`` once (sailh.py:348); SAILH then uses
``canopy.lidf`` and never looks at LIDFa / LIDFb again.  So assigning ``lidf`` (or editing it in place) and ``nlayers`` changes
its answer, while editing ``LIDFa`` after construction does not; the fixtures pin all of that.
"""
import numpy as np

# a leaf inclination distribution of this repository's own making over the 13 classes of sailh.py:49 (centres 5 ... 75, 81 ... 89
# degrees): mostly horizontal leaves.  Sums to 1 exactly in float64.
TABLE_LIDF = np.array([0.22, 0.19, 0.16, 0.13, 0.10, 0.07, 0.05, 0.03, 0.02, 0.01, 0.01, 0.005, 0.005])
OTHER_AB = (0.3, -0.2)      # "a lidf taken from another (a, b)": the generator stores calculate_leafangles(*OTHER_AB) as `payload`


def _uniform(cs, payload=None):
    cs.lidf = np.full((13, 1), 1.0 / 13.0)


def _table(cs, payload=None):
    cs.lidf = TABLE_LIDF[:, None].copy()


def _flat13(cs, payload=None):
    cs.lidf = TABLE_LIDF.copy()                   # a 1-D (13,) array instead of the constructor's (13, 1) column


def _other_ab(cs, payload=None):
    cs.lidf = np.array(payload, dtype=np.float64).reshape(13, 1)


def _inplace(cs, payload=None):
    li = cs.lidf                                  # the array SAILH will read, edited where it lives
    li[0] += 0.05
    li[12] -= 0.05


def _unnormalised(cs, payload=None):
    cs.lidf = 1.25 * TABLE_LIDF[:, None]          # sums to 1.25: the reference's dot products take it as it is (sailh.py:93-97)


def _nlayers(n):
    def edit(cs, payload=None):
        cs.nlayers = n
    return edit


def _lidfa_after(cs, payload=None):
    cs.LIDFa = 0.4                                # after construction: lidf keeps the constructor's (a, b) (sailh.py:348)
    cs.LIDFb = 0.1


def _both(cs, payload=None):
    cs.lidf = TABLE_LIDF[:, None].copy()
    cs.nlayers = 24


# name -> edit(canopy, payload) of an object with .lidf / .nlayers / .LIDFa / .LIDFb
CANOPY_EDITS = {"uniform": _uniform, "table": _table, "flat13": _flat13, "other_ab": _other_ab, "inplace": _inplace,
                "unnormalised": _unnormalised, "nlayers30": _nlayers(30), "nlayers120": _nlayers(120), "nlayers7": _nlayers(7),
                "nlayers1": _nlayers(1), "lidfa_after": _lidfa_after, "table_nlayers24": _both}
# what the edit hands to a kernel-level call (tests through the engine / C ABI): nlayers of each edit (None = 60)
NLAYERS_OF = {"nlayers30": 30, "nlayers120": 120, "nlayers7": 7, "nlayers1": 1, "table_nlayers24": 24}
