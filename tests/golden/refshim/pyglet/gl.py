"""empty stand-in"""
