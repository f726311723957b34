"""Drop-in namespace: the reference's dotted paths (yaml `target:` strings, script imports) resolve to the
MI355X-native classes in adaface_amd.ldm.  Thin re-exports only — no logic lives here."""
