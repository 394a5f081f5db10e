"""File-reader harness: this repository's counterpart of the reference's run_text_to_file_reader.py (:8-16 read_texts,
:19-41 the_raven, :44-51 main) on the MI355X-native ToucanTTSInterface, driven by phoneme strings.

Same functions, same arguments, same output file (audios/the_raven_<version>.wav, 24 kHz, 10 600 samples of silence around
every sentence).  The reference hands raw text to espeak-ng; this repository has no grapheme-to-phoneme step (raw text raises
in ``ToucanTTSInterface``), so the fourteen lines of the poem are carried here as phoneme strings (IPA, General American, the
symbol set of the articulatory table) and ``the_raven`` calls ``read_texts(..., input_is_phones=True)``.

    python run_phoneme_file_reader.py [--models-dir DIR] [--fixture-weights] [--avocodo]

--fixture-weights writes the seeded fixture checkpoints (reference file layout, random-init "tamed" weights) into the models
directory first - there is no network for the real ones (run_model_downloader.py); the audio is then noise-like but the whole
path (checkpoint load, phoneme front end, batched synthesis, file writer) is the real one.
"""
import argparse
import os

import torch

from InferenceInterfaces.ToucanTTSInterface import ToucanTTSInterface

# the first fourteen lines of the poem the reference's script reads (plain text, for the record: what THE_RAVEN_PHONES transcribes)
THE_RAVEN = [line.strip() for line in """
    Once upon a midnight dreary, while I pondered, weak, and weary,
    Over many a quaint, and curious volume of forgotten lore,
    While I nodded, nearly napping, suddenly, there came a tapping,
    As of someone gently rapping, rapping at my chamber door.
    Tis some visitor, I muttered, tapping at my chamber door,
    Only this, and nothing more.
    Ah, distinctly, I remember, it was in the bleak December,
    And each separate dying ember, wrought its ghost upon the floor.
    Eagerly, I wished the morrow, vainly, I had sought to borrow
    From my books surcease of sorrow, sorrow, for the lost Lenore,
    For the rare and radiant maiden, whom the angels name Lenore,
    Nameless here, for evermore.
    And the silken, sad, uncertain, rustling of each purple curtain
    Thrilled me, filled me, with fantastic terrors, never felt before.
""".strip().splitlines()]

# the same lines as phoneme strings ('~' pause, '#' end of utterance, 'ˈ' primary stress on the following vowel, 'ː' length)
THE_RAVEN_PHONES = ['~wˈʌns əpˈɑːn ɐ mˈɪdnaɪt dɹˈɪɹi~ wˈaɪl aɪ pˈɑːndɚd~ wˈiːk~ ænd wˈɪɹi~#',
                    '~ˈoʊvɚ mˈɛni ɐ kwˈeɪnt~ ænd kjˈʊɹiəs vˈɑːljuːm ʌv fɚɡˈɑːtən lˈoːɹ~#',
                    '~wˈaɪl aɪ nˈɑːdᵻd~ nˈɪɹli nˈæpɪŋ~ sˈʌdənli~ ðɛɹ kˈeɪm ɐ tˈæpɪŋ~#',
                    '~æz ʌv sˈʌmwʌn dʒˈɛntli ɹˈæpɪŋ~ ɹˈæpɪŋ æt maɪ tʃˈeɪmbɚ dˈoːɹ.~#',
                    '~tˈɪz sʌm vˈɪzɪɾɚ~ aɪ mˈʌɾɚd~ tˈæpɪŋ æt maɪ tʃˈeɪmbɚ dˈoːɹ~#',
                    '~ˈoʊnli ðˈɪs~ ænd nˈʌθɪŋ mˈoːɹ.~#',
                    '~ˈɑː~ dɪstˈɪŋktli~ aɪ ɹᵻmˈɛmbɚ~ ɪt wʌz ɪnðə blˈiːk dᵻsˈɛmbɚ~#',
                    '~ænd ˈiːtʃ sˈɛpɹət dˈaɪɪŋ ˈɛmbɚ~ ɹˈɔːt ɪts ɡˈoʊst əpˈɑːn ðə flˈoːɹ.~#',
                    '~ˈiːɡɚli~ aɪ wˈɪʃt ðə mˈɑːɹoʊ~ vˈeɪnli~ aɪ hæd sˈɔːt tə bˈɑːɹoʊ~#',
                    '~fɹʌm maɪ bˈʊks sɚsˈiːs ʌv sˈɑːɹoʊ~ sˈɑːɹoʊ~ fɔːɹ ðə lˈɔst lənˈoːɹ~#',
                    '~fɔːɹ ðə ɹˈɛɹ ænd ɹˈeɪdiənt mˈeɪdən~ hˈuːm ðɪ ˈeɪndʒəlz nˈeɪm lənˈoːɹ~#',
                    '~nˈeɪmləs hˈɪɹ~ fɔːɹ ɛvɚmˈoːɹ.~#',
                    '~ænd ðə sˈɪlkən~ sˈæd~ ʌnsˈɜːtən~ ɹˈʌslɪŋ ʌv ˈiːtʃ pˈɜːpəl kˈɜːtən~#',
                    '~θɹˈɪld miː~ fˈɪld miː~ wɪð fæntˈæstɪk tˈɛɹɚz~ nˈɛvɚ fˈɛlt bᵻfˈoːɹ.~#']


def read_texts(model_id, sentence, filename, device="cpu", language="en", speaker_reference=None, faster_vocoder=False, input_is_phones=False):
    """One interface, one language, optionally one reference voice, any number of sentences into one file (reference :8-16)."""
    sentences = [sentence] if isinstance(sentence, str) else list(sentence)
    tts = ToucanTTSInterface(device=device, tts_model_path=model_id, faster_vocoder=faster_vocoder)
    tts.set_language(language)
    if speaker_reference is not None:
        tts.set_utterance_embedding(speaker_reference)
    tts.read_to_file(text_list=sentences, file_location=filename, input_is_phones=input_is_phones)


def the_raven(version, model_id="Meta", exec_device="cpu", speed_over_quality=True, speaker_reference=None):
    """audios/the_raven_<version>.wav (reference :19-41), from the phoneme strings."""
    os.makedirs("audios", exist_ok=True)
    read_texts(model_id, THE_RAVEN_PHONES, os.path.join("audios", f"the_raven_{version}.wav"), device=exec_device,
               language="en", speaker_reference=speaker_reference, faster_vocoder=speed_over_quality, input_is_phones=True)


if __name__ == '__main__':
    ap = argparse.ArgumentParser()
    ap.add_argument("--models-dir", default=None, help="where Models/ToucanTTS_Meta/best.pt etc. live (default: Models/)")
    ap.add_argument("--fixture-weights", action="store_true", help="write the seeded fixture checkpoints there first")
    ap.add_argument("--avocodo", action="store_true", help="speed over quality even on a GPU (the reference picks BigVGAN there)")
    args = ap.parse_args()
    from ims_toucan_prosody_variance_amd import interface
    if args.models_dir:
        interface.MODELS_DIR = args.models_dir
    if args.fixture_weights:
        interface.write_fixture_checkpoints(interface.MODELS_DIR)
    exec_device = "cuda" if torch.cuda.is_available() else "cpu"
    print(f"running on {exec_device}")
    the_raven(version="MetaBaseline",
              model_id="Meta",
              exec_device=exec_device,
              speed_over_quality=args.avocodo or exec_device != "cuda")
