"""Drop-in for the reference package `voicebox` (src/decoder/voicebox): same module paths, class and
function names, constructor arguments and state-dict keys; the arithmetic runs in libusdm_hip.so."""
