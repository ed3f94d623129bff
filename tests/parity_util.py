"""The tolerance rules live beside the oracle (oracle/parity_util.py) so that bench.py's parity gate applies the SAME rules as the tests."""
from oracle.parity_util import *  # noqa: F401,F403
from oracle.parity_util import G_TOL, NET_FLOOR, P_ROUND  # noqa: F401
