"""MI355X-native KEEP inference engine (host side).  See DESIGN.md."""
