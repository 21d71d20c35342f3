"""MI355X-native lattice-Boltzmann stream-and-collide engine with lettuce's Python API."""
