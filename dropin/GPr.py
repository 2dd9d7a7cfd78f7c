"""Put this directory on sys.path so the reference's demo scripts resolve ``import GPr``
to the MI355X-native implementation:  PYTHONPATH=/root/repo:/root/repo/dropin python GP_regression_demo.py"""
from gptest_amd.GPr import *  # noqa: F401,F403
from gptest_amd.GPr import squared_distance, GaussianProcess, SquaredExponential  # noqa: F401
