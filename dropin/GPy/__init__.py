"""``import GPy`` for the reference's scripts (GP_parameter_fit.py:1,30-33, GPpref.py:3,109): the few GPy calls they make,
served by the MI355X core (gptest_amd.gpy_compat).  Not a GPy re-implementation."""
from gptest_amd import gpy_compat as _c


class kern(object):
    RBF = _c.RBF


class models(object):
    GPRegression = _c.GPRegression
