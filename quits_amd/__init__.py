"""quits_amd -- MI355X-native sliding-window BP-OSD decoder behind the `quits.decoder` API.

Only the decoding hot path of mkangquantum/quits is implemented here:
code construction, circuit generation and Stim sampling stay upstream.
"""

__version__ = "0.1.0"
