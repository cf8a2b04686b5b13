"""MI355X-native self-play engine behind the reference's Game / MCTS interfaces.

Put this directory on sys.path to get the reference's import names (`ai`, `games`) plus
`azk` (ctypes binding of libazk.so) and `selfplay` (batched self-play driver).
"""
