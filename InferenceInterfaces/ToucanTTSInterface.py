"""Same import path as the reference (InferenceInterfaces/ToucanTTSInterface.py:21): scripts written against the
reference, e.g. run_text_to_file_reader.py:3 ``from InferenceInterfaces.ToucanTTSInterface import ToucanTTSInterface``,
pick up the MI355X-native implementation when this repository is on sys.path."""
import ims_toucan_prosody_variance_amd  # noqa: F401
from ims_toucan_prosody_variance_amd.interface import ToucanTTSInterface  # noqa: F401
