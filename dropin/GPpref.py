"""``import GPpref`` -> the MI355X-native implementation (see dropin/GPr.py)."""
from gptest_amd.GPpref import *  # noqa: F401,F403
from gptest_amd.GPpref import PrefProbit, PreferenceGaussianProcess, squared_distance, std_norm_pdf  # noqa: F401
