"""Golden vectors for the phoneme-string front end.  Runs ONLY where /root/reference exists.

Feeds phoneme strings through the reference's own ``ArticulatoryCombinedTextFrontend.string_to_tensor(..., input_phonemes=True)``
(Preprocessing/TextFrontend.py:213-288) and ``get_language_id`` (:490-524) and stores strings + expected outputs as
``tests/golden/frontend.json`` (data only).  ``tests/test_frontend_golden.py`` replays them against
``ims-toucan-prosody-variance_amd/phonemes.py``.

    python tests/golden/make_frontend_golden.py
"""
import contextlib
import io
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden  # noqa: E402,F401  (installs the stand-in modules and the sys.path entries)

from Preprocessing.TextFrontend import ArticulatoryCombinedTextFrontend, get_language_id  # noqa: E402

STRINGS = [
    "~həlˈoʊ wˈɜːld~#",
    "~ˈaɪ sˈi tˈu~#",
    "ˈaː˥ ñ",
    "maˑ˦ tĕ˧ lo˨ ku˩",
    "pa⭧ ti⭨ ko⮁ bu⮃",
    "ðɪs ɪz ɐ tˈɛst, wɪð pˈɔːzᵻz? jˈɛs! ənd ɚ dˈɑːt.",
    "ʃtʁˈaːsə ˈʏbɐ ɡəmˈyːtlɪç~#",
    "xɤ˧˥ ʈʂʰɤŋ˥˩ ni˨˩˦",
    "a§b€c",                      # unknown symbols are skipped (handle_missing=True)
    "~#",
    "ɡ g ʔ ǀ ǁ ǂ ǃ ʘ",
    "a\u0303 e\u0306 oː\u0303",      # combining tilde / breve (decomposed forms are the ones the table knows)
    "aˈ§bˈi",                     # a stress mark followed by an unknown symbol: the stress lands on the PREVIOUS phoneme (:282-286)
    "tˈ€ˈ$oː",                    # twice in a row, then a modifier on the following known phoneme
]
LANGS = ["de", "el", "es", "fi", "ru", "hu", "nl", "fr", "pt", "pl", "it", "en", "cmn", "vi", "uk", "fa", "pt-br", "xx"]


def main():
    with contextlib.redirect_stdout(io.StringIO()):
        tf = ArticulatoryCombinedTextFrontend(language="en")
    cases = []
    for s in STRINGS:
        buf = io.StringIO()
        with contextlib.redirect_stdout(buf):
            t = tf.string_to_tensor(s, input_phonemes=True)
        rows = ["".join(str(int(v)) for v in r) for r in t.tolist()] if t.dim() == 2 else []
        cases.append({"phones": s, "rows": rows, "printed": buf.getvalue()})
    langs = {}
    for lang in LANGS:
        with contextlib.redirect_stdout(io.StringIO()):
            try:
                v = get_language_id(lang)
            except SystemExit:
                v = "exit"
        langs[lang] = None if v is None else (v if isinstance(v, str) else int(v.item()))
    out = os.path.join(HERE, "frontend.json")
    with open(out, "w", encoding="utf-8") as f:
        json.dump({"cases": cases, "language_ids": langs}, f, ensure_ascii=False, indent=1)
    print("wrote", out, sum(len(c["rows"]) for c in cases), "rows")


if __name__ == "__main__":
    main()
